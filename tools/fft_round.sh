export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "stft or istft or feature or separate or loader or label or roundtrip" 2>&1 | tail -3
cp onssen_amd/libonssen_hip.so /tmp/keep.so
cp build_variants/libonssen_hip_knobs.so onssen_amd/libonssen_hip.so
for ppw in 1 2 4 8; do ONSSEN_STFT_PPW=$ppw ONSSEN_ISTFT_FB=16 timeout 120 python tools/fft_probe.py 2>&1 | tail -3; done
for fb in 8 12; do ONSSEN_STFT_PPW=4 ONSSEN_ISTFT_FB=$fb timeout 120 python tools/fft_probe.py 2>&1 | tail -3; done
cp /tmp/keep.so onssen_amd/libonssen_hip.so
timeout 120 python tools/fft_probe.py 2>&1 | tail -3
timeout 900 python tools/ab_variants.py run base cnt -- bench.py --no-cpu-baseline --no-extra --steps 40 2>&1 | tail -8
