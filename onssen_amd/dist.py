"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests).

Inference shards *utterances*: a BLSTM cannot be split along time (the reverse direction needs the
whole chunk) and splitting the hidden units would put a collective inside every one of the T serial
steps, while utterances are independent -- so the data path has no collective at all.
Training is data parallel with one exchange per optimizer step: a bucketed all-reduce of the
gradients placed between ``backward()`` and ``clip_grad_norm_`` (onssen/utils/train.py:82-83), so
that clipping sees the global gradient.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice of ``n_items`` utterances for ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_utterances(local, n_items, world, group=None):
    """Concatenate per-rank result tensors (dim 0 = utterances of ``shard_range``) on every rank."""
    if world == 1:
        return local
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)


def gradient_buckets(model):
    """Parameters grouped so that a bucket becomes ready as one unit during backward: one bucket per
    LSTM (layer, direction) -- reversed registration order ~ the order gradients are produced -- and one
    for everything else (heads, BatchNorm).  ~11.5-23 MB each at H=600: large enough for xGMI's
    per-link bandwidth, small enough to overlap with the remaining backward."""
    lstm, rest = {}, []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        key = None
        if "weight_ih_l" in name or "weight_hh_l" in name or "bias_ih_l" in name or "bias_hh_l" in name:
            key = name.split(".")[:-1] + [name.split("_l")[-1]]          # module path + "<k>[_reverse]"
            key = ".".join(key)
        if key is None:
            rest.append(p)
        else:
            lstm.setdefault(key, []).append(p)
    buckets = [ps for _, ps in sorted(lstm.items(), reverse=True)]
    if rest:
        buckets.insert(0, rest)
    return buckets


def allreduce_gradients(model, world, group=None, buckets=None):
    """Average gradients across ranks: one flat fp32 all-reduce(sum) per bucket, issued
    asynchronously, then scaled by 1/world and scattered back."""
    if world == 1:
        return
    buckets = buckets if buckets is not None else gradient_buckets(model)
    pending = []
    for ps in buckets:
        ps = [p for p in ps if p.grad is not None]
        if not ps:
            continue
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, ps))
    for work, flat, ps in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def train_step(model, optimizer, loss_fn, input, label, world=1, group=None, clip_norm=5.0):
    """One optimizer step in the order of onssen/utils/train.py:75-86 with the data-parallel
    exchange inserted before gradient clipping.  Returns the local mean loss (float)."""
    output = model(input)
    loss_avg = torch.mean(loss_fn(output, label))
    optimizer.zero_grad()
    loss_avg.backward()
    allreduce_gradients(model, world, group)
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
    optimizer.step()
    return float(loss_avg.item())
