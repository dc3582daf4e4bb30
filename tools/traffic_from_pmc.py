#!/usr/bin/env python
"""HBM traffic per launch of the dominant kernel from this round's rocprofv3 --pmc passes (tools/profile_round.sh):
FETCH_SIZE x 2 (gfx950: 16 B/lane coalesced reads are tallied at half, MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both in
KB, averaged over the dispatches of the plain (layer >= 1) stacked recurrence launch.  Updates the "latest" block of
profiles/traffic.json, which bench.py's roofline.traffic reads -- the number then names the round it was measured in.
  python tools/traffic_from_pmc.py gpurun_out/prof_<tag> <tag>"""
import collections, csv, glob, json, os, sys

root, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{root}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r.get("Kernel_Name", "")][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))


def mean(name_part, counter, exclude=None):
    v = [x for k, cs in acc.items() if name_part in k and not (exclude and exclude in k) for x in cs.get(counter, [])]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


out = {"round": tag, "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1 "
                               f"--no-graph (dc_l2, 32 x 400 frames); tools/profile_round.sh {tag}"}
# the plain stacked instantiation (FUSE = false, TERMS = 3, STACK = true): layer 1 of the headline step
kname = "lstm_xcd_kernel<5, 8, false, 3, true"
f, nf = mean(kname, "FETCH_SIZE")
w, nw = mean(kname, "WRITE_SIZE")
if f is not None and w is not None:
    out.update({"kernel": kname + ", ...> (layer >= 1 of the headline step)", "xcd_FETCH_SIZE_KB_per_launch_raw": f,
                "xcd_WRITE_SIZE_KB_per_launch_raw": w, "dispatches": [nf, nw],
                "xcd_recurrence_hbm_bytes_per_launch": int((2 * f + w) * 1024)})
# the PAIR launch of the pipelined step (onssen_blstm_pipe2_forward_f32): the unstacked instantiation on 16-row groups, two layers' worth
pname = "lstm_xcd_kernel<5, 8, false, 3, false, false"
f, nf = mean(pname, "FETCH_SIZE")
w, nw = mean(pname, "WRITE_SIZE")
if f is not None and w is not None:
    out.update({"pair_kernel": pname + ", ...> (layer 1 of batch k-1 || layer 0 of batch k)", "xcd_pair_FETCH_SIZE_KB_per_launch_raw": f,
                "xcd_pair_WRITE_SIZE_KB_per_launch_raw": w, "pair_dispatches": [nf, nw],
                "xcd_pair_recurrence_hbm_bytes_per_launch": int((2 * f + w) * 1024),
                "xcd_pair_algorithmic_bytes_per_launch": 2 * 319539200})
for label, part in (("linear_x3q_l1", "linear_x3q_kernel<0, 3, false, 3"), ("linear_x3q_head_full_embedding", "linear_x3q_kernel<1, 3"),
                    ("linear_x3q_head_active_rows_only", "linear_x3q_kernel<4, 3"), ("kmeans2_search_farthest", "kmeans2_search_kernel<2"),
                    ("kmeans2_lloyd", "kmeans2_lloyd_kernel"),
                    ("kmeans2_compact", "kmeans2_compact_kernel"), ("kmeans2_assign_masks", "kmeans2_assign_kernel<1"),
                    ("kmeans2_mask_compact", "kmeans2_mask_compact_kernel"), ("kmeans2_index", "kmeans2_index_kernel")):
    f, _ = mean(part, "FETCH_SIZE")
    w, _ = mean(part, "WRITE_SIZE")
    if f is not None:
        out[label] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "hbm_bytes": int((2 * f + (w or 0)) * 1024)}
print(json.dumps(out, indent=1))
tf = os.path.join(ROOT, "profiles", "traffic.json")
if "xcd_recurrence_hbm_bytes_per_launch" in out and os.path.exists(tf):
    tj = json.load(open(tf))
    tj["latest"] = out
    # written next to the profile too: gpurun only brings gpurun_out/ back, the copy into profiles/ happens in the build container
    json.dump(tj, open(os.path.join(root, "traffic.json"), "w"), indent=1)
