"""Batch-shard data parallelism over the 8 GPUs of one node.

One process per GPU (torchrun); every image is independent end to end, so the only exchange is ONE
all-gather of the feature matrix into its final row order (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Pure gather, no reduction: the N-rank result is bitwise equal to
the 1-rank result.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(script: str, nproc: int, args=(), env=None, timeout=None) -> int:
    """Start `nproc` fresh rank processes of `script` on this node (one per GPU) and wait for them:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script args`.
    The ranks are CHILD processes created with subprocess (fork + exec of a NEW interpreter: the caller is never replaced
    and may itself have initialised the GPU -- nothing of its state reaches the children).  The launcher runs in its own
    session: on `timeout` the whole process group (launcher + ranks) gets SIGTERM, then SIGKILL, so no rank is left
    behind holding a GPU.  stdout / stderr of the ranks pass through.  Returns the launcher's exit code (0 = every rank
    exited 0; 124 after a timeout)."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)  # a stale rendezvous in the caller's environment must not leak into the children
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *[str(a) for a in args]]
    proc = subprocess.Popen(cmd, env=e, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def init_from_env(backend: str = "nccl", device=None):
    """Join the process group torch.distributed.run (or launch_ranks) prepared: RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* from the environment.  backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors.
    Returns (rank, world, local_rank)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=device if device is not None else torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; the first total % world ranks get one extra row."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """Gather per-rank row blocks [rows_r, ...] into [total_rows, ...] in rank (== dataset) order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    bounds = [shard_bounds(total_rows, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in bounds]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    out = torch.empty((total_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # one in-place collective
    else:
        # ragged tail: pad every shard to the largest, one collective, then drop the padding rows
        mx = max(sizes)
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: sizes[rank]] = local
        tmp = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(tmp, pad, group=group)
        for r, (lo, hi) in enumerate(bounds):
            out[lo:hi] = tmp[r * mx: r * mx + (hi - lo)]
    return out


def extract_features_sharded(extract_fn, images_local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """extract_fn(images_local) -> [rows_local, D] on this rank's device; returns the full [N, D]
    matrix on every rank (the array umap_cls_token.py:139 stacks / embedding_classifier.py:102 reads)."""
    return all_gather_rows(extract_fn(images_local), total_rows, group)


# ---- the step that hides its gather (reference consumers: umap_cls_token.py:133-139) ------------------------
MESH_THRESHOLD_BYTES = 8 << 20  # per-rank message above which "auto" takes the full-mesh point-to-point path


def gather_plan(rows_local_bytes: int, chunks: int = 0, mode: str = "auto"):
    """(chunks, mode) for a per-rank message of that many bytes: small messages (the [B, D] CLS matrix: < 1 MB per
    rank) go out as ONE collective after the forward -- splitting the forward would cost more than the gather takes;
    large ones (dense per-patch descriptors: 75 MB per rank and step at BASELINE config 4) are cut in 4 micro-batches
    whose gathers run on a side stream under the next micro-batch's kernels, point to point over the full xGMI mesh.
    The caller passes a RANK-INVARIANT size (the largest shard): every rank must arrive at the same plan."""
    big = rows_local_bytes > MESH_THRESHOLD_BYTES
    if chunks <= 0:
        chunks = 4 if big else 1
    if mode == "auto":
        mode = "mesh" if big else "collective"
    if mode not in ("mesh", "collective"):
        raise ValueError(f"gather mode {mode!r}: 'auto', 'mesh' or 'collective'")
    return chunks, mode


class GroupTransport:
    """The peers are the ranks of a torch.distributed process group.

    mode "collective": all_gather_into_tensor per micro-batch into a [world, rows, ...] staging buffer + one strided
        device copy into the peers' row slices (RCCL picks the algorithm; equal shards only);
    mode "mesh": every rank posts, per micro-batch, one send of its rows to each peer and one receive from each peer
        DIRECTLY into that peer's row slice of the final matrix (batch_isend_irecv: RCCL runs the 2 (world - 1) transfers
        of a group concurrently, one per xGMI link of the point-to-point mesh; no staging, ragged shards welcome).

    `stream_ordered`: only RCCL ("nccl") enqueues its transfers on the CUDA stream that is current when they are posted.
    gloo hands a device pointer to its TCP transport -- the host reads VRAM with no ordering against any stream -- so for
    device tensors on any other backend the owner synchronises the compute stream before every post and uses no side
    stream."""

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.stream_ordered = dist.get_backend(group) == "nccl"
        self._work, self._stage = [], None

    def post(self, og, c: int):
        feats = og.feats
        if og.mode == "mesh":
            ops = []
            for k in range(1, self.world):  # staggered partners: every rank talks to a different peer in round k
                dst, src = (self.rank + k) % self.world, (self.rank - k) % self.world
                a, b = og.chunk_bounds(self.rank, c)
                if b > a:
                    ops.append(dist.P2POp(dist.isend, feats[a:b], dst, self.group))
                a, b = og.chunk_bounds(src, c)
                if b > a:
                    ops.append(dist.P2POp(dist.irecv, feats[a:b], src, self.group))
            if ops:
                self._work += dist.batch_isend_irecv(ops)
            return
        a, b = og.chunk_bounds(self.rank, c)
        n = b - a
        if n == 0:
            return
        if og.chunks == 1:  # the whole shard at once: the in-place collective (input = this rank's slice of the output)
            dist.all_gather_into_tensor(feats, feats[a:b], group=self.group, async_op=True).wait()
            return
        tail = tuple(feats.shape[1:])
        if self._stage is None:
            mx = max(og.chunk_bounds(self.rank, cc)[1] - og.chunk_bounds(self.rank, cc)[0] for cc in range(og.chunks))
            self._stage = torch.empty((self.world, mx) + tail, dtype=feats.dtype, device=feats.device)
        stage = self._stage if self._stage.shape[1] == n else None
        if stage is None:  # (the last micro-batch of an uneven split: a contiguous staging buffer of its own size)
            stage = torch.empty((self.world, n) + tail, dtype=feats.dtype, device=feats.device)
        # .wait() orders the current (side) stream behind the collective; the compute stream is not involved
        dist.all_gather_into_tensor(stage.view((self.world * n,) + tail), feats[a:b], group=self.group, async_op=True).wait()
        for r in range(self.world):
            if r != self.rank:
                ra, rb = og.chunk_bounds(r, c)
                feats[ra:rb].copy_(stage[r])

    def finish(self):
        for w in self._work:
            w.wait()
        self._work = []


class LoopbackTransport:
    """ONE process plays `world` ranks on one device: the peers' rows come from `peer_rows` (a complete [N, ...] matrix,
    e.g. the whole-batch forward) by device-to-device copies, and what this rank "sends" is copied into `sent` -- all
    enqueued on the stream that is current at post(), i.e. OverlappedGather's side stream.  It exists so that the event /
    side-stream / finish() wiring of the overlapped step runs on a device where only one GPU is visible
    (tests/test_dist_gpu.py): a post that is not ordered behind its micro-batch's forward shows up as stale rows in `sent`."""
    stream_ordered = True

    def __init__(self, world: int, rank: int, peer_rows: torch.Tensor):
        assert 0 <= rank < world
        self.world, self.rank, self.peer = world, rank, peer_rows
        self.sent = torch.full_like(peer_rows, float("nan")) if peer_rows.is_floating_point() else torch.zeros_like(peer_rows)
        self.posts = []

    def post(self, og, c: int):
        assert og.feats.shape == self.peer.shape
        a, b = og.chunk_bounds(self.rank, c)
        if b > a:
            self.sent[a:b].copy_(og.feats[a:b], non_blocking=True)       # the "isend" of this micro-batch
        for r in range(self.world):
            if r != self.rank:
                ra, rb = og.chunk_bounds(r, c)
                if rb > ra:
                    og.feats[ra:rb].copy_(self.peer[ra:rb], non_blocking=True)  # the "irecv" from peer r
        self.posts.append((c, torch.cuda.current_stream(og.feats.device).cuda_stream if og.feats.is_cuda else None))

    def finish(self):
        pass


class OverlappedGather:
    """One data-parallel extraction step: this rank's forward writes its rows STRAIGHT into its slice of the final
    row-ordered [N, ...] matrix, micro-batch by micro-batch, and every finished micro-batch is sent to the peers while the
    next one computes.  The transfers belong to a transport (`post(og, c)`, `finish()`): GroupTransport (RCCL / gloo
    process group, "mesh" or "collective") or LoopbackTransport (one process playing every rank).  Pure copies either way:
    the matrix is bitwise the 1-rank result.

    With device tensors and a stream-ordered transport the transfers are issued from a side stream that waits for the
    micro-batch's forward only (an event), so the compute stream never waits for a transfer until `finish()`; with a
    transport that ignores streams (gloo on device tensors) the compute stream is synchronised before every post; on CPU
    tensors (gloo, the tests) the same calls run in order.  `gather_ms` (GPU) is the time the transfers of the last step
    occupied the side stream."""

    def __init__(self, feats: torch.Tensor, total_rows: int, chunks: int = 0, mode: str = "auto", group=None, transport=None):
        self.transport = transport if transport is not None else GroupTransport(group)
        self.world, self.rank = self.transport.world, self.transport.rank
        assert feats.shape[0] == total_rows and feats.is_contiguous()
        self.feats, self.total = feats, total_rows
        self.bounds = [shard_bounds(total_rows, r, self.world) for r in range(self.world)]
        row_bytes = feats[0].numel() * feats.element_size() if total_rows else 0
        largest = max(b - a for a, b in self.bounds)
        # planned from the LARGEST shard, which every rank computes alike: ranks whose own ragged shards straddle
        # MESH_THRESHOLD_BYTES would otherwise post different numbers of transfers per step
        self.chunks, self.mode = gather_plan(largest * row_bytes, chunks, mode)
        self.chunks = max(1, min(self.chunks, largest or 1))
        sizes = {b - a for a, b in self.bounds}
        if self.mode == "collective" and len(sizes) != 1:
            self.mode = "mesh"  # ragged shards: the point-to-point path needs no padding
        self.cuda = feats.is_cuda
        self.stream_ordered = bool(getattr(self.transport, "stream_ordered", False))
        self.side = torch.cuda.Stream(device=feats.device) if (self.cuda and self.world > 1 and self.stream_ordered) else None
        self._events = []
        self.gather_ms = None

    def my_rows(self):
        return self.bounds[self.rank]

    def chunk_bounds(self, r: int, c: int):
        """rows [lo, hi) of the final matrix that are micro-batch c of rank r"""
        lo, hi = self.bounds[r]
        a, b = shard_bounds(hi - lo, c, self.chunks)
        return lo + a, lo + b

    def run(self, forward_rows) -> torch.Tensor:
        """forward_rows(a, b, out) computes this rank's local images [a, b) into `out` (= their rows of the final
        matrix, a view); returns the complete matrix."""
        lo, _ = self.my_rows()
        t0 = t1 = None
        for c in range(self.chunks):
            a, b = self.chunk_bounds(self.rank, c)
            if b > a:
                forward_rows(a - lo, b - lo, self.feats[a:b])
            if self.world == 1:
                continue
            if self.side is not None:
                ev = torch.cuda.Event()
                ev.record()
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    if c == 0:
                        t0 = torch.cuda.Event(enable_timing=True)
                        t0.record()
                    self.transport.post(self, c)
                    if c == self.chunks - 1:
                        t1 = torch.cuda.Event(enable_timing=True)
                        t1.record()
            else:
                if self.cuda:  # a transport that does not follow streams reads the rows from the host side: they must be there
                    torch.cuda.current_stream(self.feats.device).synchronize()
                self.transport.post(self, c)
        self.finish()
        if t0 is not None and t1 is not None:
            self._events = [(t0, t1)]
        return self.feats

    def finish(self):
        self.transport.finish()
        if self.side is not None:
            torch.cuda.current_stream(self.feats.device).wait_stream(self.side)

    def last_gather_ms(self):
        """time between the first transfer's start and the last transfer's end on the side stream (call after a device
        synchronisation); None on CPU / world 1"""
        if not self._events:
            return None
        t0, t1 = self._events[0]
        return t0.elapsed_time(t1)
