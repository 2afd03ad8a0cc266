"""Throughput runner: overlap the serial sampling stage with the dense stages.

Furthest point sampling is one workgroup per cloud -- 2B of the 256 CUs for a batch of B scan pairs --
and takes about as long as all other stages together, which in turn cannot start without its result.
For a STREAM of batches the dependency is only within a batch, so the runner issues sampling for
batches i+1 .. i+depth on side HIP streams while set abstraction, flow embedding and the pose head
of batch i run on the main stream (the reference never batches or pipelines: one pair per call,
/root/reference/deepclr/models/base.py:118-120, scripts/inference.py:100-104).
"""
import os
from collections import deque
from typing import Deque, Iterable, Iterator, Optional, Tuple

import torch

from . import ops
from .models.deepclr import DeepCLR


class PipelinedForward:
    def __init__(self, model: DeepCLR, depth: int = 3, ahead: str = 'features', group: int = 1,
                 dense_group: bool = False, inputs_ready: bool = False, dense_streams: int = 1,
                 eager_dense: Optional[bool] = None):
        """ahead: what runs on the side streams -- 'sample' (sampling only), 'features' (sampling + set
        abstraction; the dense kernels of two batches then overlap and fill each other's tails) or 'knn' (also
        the kNN search and the per-point halves of flow layer 1, which need nothing but the feature rows; the
        main stream is then left with the flow embedding, the head and the fully connected tail).
        group: batches sampled by ONE launch (not with ahead='sample'). The sampler is a latency chain (~1 ms per
        launch, one workgroup per cloud), so its throughput is launches in flight x clouds per launch; the HIP
        runtime multiplexes streams onto 4 hardware queues by default (GPU_MAX_HW_QUEUES; bench.py raises it to
        8 so that RCCL's own stream does not share a queue with a sampling launch) and more than 3 side streams
        measured slower either way -- grouping is how more clouds get in flight (their inputs are concatenated
        on the side stream).
        dense_group (needs ahead='knn'): the dense stages of the `group` batches sampled together also run as ONE
        launch sequence (flow embedding, head and fully connected tail over group x B pairs, enqueued when the
        first batch of the group is stepped): a single batch of 8 KITTI pairs is 8192 head rows = 128 workgroups,
        half the chip, and three ~13 us fully connected launches per batch were a fifth of the main stream.
        inputs_ready: the batches handed in are complete in memory and stay untouched until their results are out
        (resident data, or produced on another stream and already synchronised). By default a sampling launch waits
        for everything enqueued on the caller's stream so far -- the safe assumption that the batch was produced
        there -- which also orders it behind the dense launches of OLDER batches the runner itself put on that
        stream; with inputs_ready it starts at once.
        dense_streams (single-batch dense launches only, i.e. not with dense_group): the dense stages of consecutive batches
        alternate over this many high-priority streams of the runner's own instead of queueing on the caller's stream. One
        batch's dense stages are six dependent launches (flow embedding, clear, head, three fully connected layers), 0.25-
        0.37 ms under contention for 8 KITTI pairs: on ONE stream that chain, not the sampler, sets the pace of the un-fused
        regime (bench.py --strict). The caller's stream waits for each batch's outputs (an event), so results are ordered
        for the caller exactly as before.
        eager_dense (dense_group + inputs_ready only; default off): the dense stages of a group are enqueued on the
        caller's stream at once, right behind the group's sampling launch (they wait for its event on the device), instead
        of at the step that hands out the group's first batch up to `group` steps later. Built in round 6 on the reading
        that a short timed window leaves the device waiting for the host to reach a group boundary -- and measured WORSE:
        the driver's 20-step window 40.7-40.9k pairs/s against 42.3-43.1k, 200 steps 46.5k against 47.2k (three
        alternating runs on one box, profiles/NOTES.md). With the dense stages enqueued ahead, a fence drains them too: a
        window then opens on an EMPTY pipeline and its first dense launch waits for a whole sampling chain, where the
        default finds groups that are sampled already and only need their dense stages. Kept as an option for callers
        that want a group's outputs as early as the device can produce them (lowest result latency). A caller-supplied
        `out` is filled by a copy of the group's outputs."""
        if dense_group and (ahead != 'knn' or group < 2):
            raise ValueError("dense_group needs ahead='knn' and group > 1")
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if dense_streams < 1 or (dense_streams > 1 and dense_group):
            raise ValueError("dense_streams > 1 applies to single-batch dense launches (dense_group=False)")
        if ahead not in ('sample', 'features', 'knn'):
            raise ValueError("ahead must be 'sample', 'features' or 'knn'")
        if group < 1 or (group > 1 and ahead == 'sample'):
            raise ValueError("group > 1 needs ahead='features' or 'knn'")
        self._model = model.eval()
        model.prepare()                             # packed weights built here, on the caller's stream
        self.depth = depth
        self.group = group
        self._ahead = ahead
        self._dense_group = dense_group
        self._inputs_ready = inputs_ready
        self._eager = bool(eager_dense)
        if self._eager and not (dense_group and inputs_ready):
            raise ValueError("eager_dense needs dense_group and inputs_ready")
        self._in_place = hasattr(model, '_cloud_layers') and os.environ.get('DCLR_BATCH_VIEW', '1') != '0'   # A/B: 0 = always concatenate
        self._planned = False                       # launch plans of the side streams built (first _launch)
        self._hold_launch = False                   # dense groups: a full sampling group is launched right AFTER the next
                                                    # dense launch has been enqueued (the host needs ~0.3 ms for the chain)
        self._group_out = None                      # (batches of the running dense group, their outputs)
        self._waiting = []                          # batches collected for the next grouped launch
        self._streams = [torch.cuda.Stream() for _ in range(depth)]
        self._dense_streams = [torch.cuda.Stream(priority=-1) for _ in range(dense_streams)] if dense_streams > 1 else []
        self._dense_next = 0
        self._next_stream = 0
        self._pending: Deque[Tuple[torch.Tensor, torch.Tensor, torch.cuda.Event]] = deque()
        self.prefetched = 0                         # batches accepted by prefetch() so far (feeders watch this)

    def prefetch(self, x: torch.Tensor, flush: bool = True, ready: Optional[torch.cuda.Event] = None) -> None:
        """Start sampling for `x` (2B, N, C) on the next side stream (with group > 1: once `group` batches have
        been handed in, or at once if flush). ready: an event after which `x` is complete in memory (e.g. recorded
        behind its host-to-device copy on a copy stream); the sampling launch waits for it -- with inputs_ready this
        is the only ordering between the batch's producer and its consumers."""
        self._waiting.append(x)
        self.prefetched += 1
        if ready is not None:
            # the event travels WITH the batch object (an attribute, gone when the tensor is): a dictionary keyed by id()
            # kept entries of batches that were never launched and handed them to whatever tensor reused the id
            x._dclr_ready = ready
        if flush or (len(self._waiting) >= self.group and not self._hold_launch):
            self._launch()

    def _launch(self) -> None:
        """One sampling (+ set abstraction) launch for every batch collected so far."""
        xs, self._waiting = self._waiting, []
        if not xs:
            return
        main = torch.cuda.current_stream()
        side = self._streams[self._next_stream]
        self._next_stream = (self._next_stream + 1) % self.depth
        if not self._inputs_ready:
            side.wait_stream(main)                           # the batches (and anything producing them) are ready
        for b in xs:
            ev = b.__dict__.pop('_dclr_ready', None)
            if ev is not None:
                side.wait_event(ev)
        fused = getattr(self._model, 'cloud_merge_prep', None) if self._ahead == 'knn' else None
        if fused is not None and not self._planned:
            # first launch: the launch plans (arguments, scratch, output ring) of EVERY side stream are built now, for this
            # launch shape -- not one by one inside whatever window the later launches fall into
            self._planned = True
            if len(xs) == 1 or self._dense_group:
                view = ops.batch_view(xs) if (len(xs) > 1 and self._in_place) else None
                if len(xs) == 1 or view is not None:
                    for s_ in self._streams:
                        with torch.cuda.stream(s_), torch.no_grad():
                            self._model.plan_cloud_forward(xs[0], view)
        with torch.cuda.stream(side), torch.no_grad():
            prepped = False
            if len(xs) == 1:
                got = fused(xs[0]) if fused is not None else None      # sampling, set abstraction, stage 1: one foreign call
                if got is not None:
                    outs, prepped = [got], True
                else:
                    out = self._model.sample(xs[0])
                    outs = [out if self._ahead == 'sample' else self._model.cloud_feature_rows(xs[0], out)]
            else:
                if any(b.shape != xs[0].shape for b in xs):
                    raise RuntimeError("batches sampled in one launch must have the same shape")
                if self._dense_group:
                    # [templates of every batch | sources of every batch]: the reference's batch layout for
                    # len(xs) * B pairs, so the dense stages can take all of them in one go
                    half = xs[0].shape[0] // 2
                    view = ops.batch_view(xs) if self._in_place else None
                    got = None
                    if view is not None:
                        # batches at a constant stride (the same resident tensor, views of one staging chunk, a ring): the
                        # sampler and set abstraction read them where they lie -- no 42 MB concatenation per ten KITTI
                        # batches -- and the whole chain is one foreign call where the model offers it
                        got = fused(xs[0], view) if fused is not None else None
                        if got is None:
                            rows = self._model.cloud_feature_rows(xs[0], self._model.sample(xs[0], view), view)
                    else:
                        big = torch.cat([b[:half] for b in xs] + [b[half:] for b in xs])
                        got = fused(big) if fused is not None else None
                        if got is None:
                            rows = self._model.cloud_feature_rows(big, self._model.sample(big))
                    if got is not None:
                        rows, prep = got
                    else:
                        prep = self._model.merge_prep(rows, half * len(xs))
                    done = torch.cuda.Event()
                    done.record(side)
                    for b in xs:
                        b.record_stream(side)
                    if not self._eager:
                        self._pending.append((xs, (rows, prep), done))
                        return
                    eager = (rows, prep, done, half * len(xs))
                if not self._dense_group:
                    big = torch.cat(xs)
                    rows = self._model.cloud_feature_rows(big, self._model.sample(big))
                    outs = list(rows.view(len(xs), -1, rows.shape[-1]).unbind(0))
            if self._dense_group and len(xs) > 1:
                pass                                             # (eager: the dense stages follow below, on the caller's stream)
            else:
                if self._ahead == 'knn' and not prepped:
                    outs = [(rows, self._model.merge_prep(rows, b.shape[0] // 2)) for b, rows in zip(xs, outs)]
                done = torch.cuda.Event()
                done.record(side)
        if self._dense_group and len(xs) > 1:
            # eager_dense: flow embedding, head and fully connected tail of the whole group, enqueued now on the caller's
            # stream behind the sampling launch's event; step() of the group's first batch finds the outputs
            rows, prep, done, want = eager
            main.wait_event(done)
            for t in self._tensors((rows, prep)):
                t.record_stream(main)
            with torch.no_grad():
                y_all = self._model.merge_rows(rows, want, prep=prep)
            if hasattr(prep, 'release'):
                prep.release()
            self._pending.append((xs, y_all, None))
            return
        for b, out in zip(xs, outs):
            b.record_stream(side)
            self._pending.append((b, out, done))

    def in_flight(self) -> int:
        running = len(self._group_out[0]) if self._group_out is not None else 0     # dense stages enqueued, not yet stepped
        return sum(len(p[0]) if isinstance(p[0], list) else 1 for p in self._pending) + len(self._waiting) + running

    def group_start(self, x: torch.Tensor) -> int:
        """Number of batches whose outputs the next step(x) will produce in one go (0: none, x continues a running
        group or is a single batch): callers that want the outputs written into a buffer of their own hand step()
        an `out` of that many batches."""
        if not self._dense_group:
            return 0
        if self._group_out is not None and self._group_out[0] and self._group_out[0][0] is x:
            return 0
        if self._pending and isinstance(self._pending[0][0], list) and self._pending[0][0][0] is x:
            return len(self._pending[0][0])
        if not self._pending and self._waiting and self._waiting[0] is x:
            return len(self._waiting)
        return 0

    def step(self, x: torch.Tensor, upcoming: Iterable[torch.Tensor] = (),
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pose outputs (B, label_dim) for batch `x` (written into `out` if given). `upcoming` lists later batches (oldest first) that are
        not yet being sampled; as many as fit the pipeline depth are started before this batch's dense
        stages are enqueued, so they run beside them."""
        ready = None
        if self._group_out is not None and self._group_out[0] and self._group_out[0][0] is x:
            # a batch of the dense group already enqueued: its outputs are a slice of that launch's result
            batches, y_all = self._group_out
            pos = y_all.shape[0] // (x.shape[0] // 2) - len(batches)
            batches.pop(0)
            self._hold_launch = not self._eager              # eager: a group that fills is launched (with its dense stages) at once
            for nxt in upcoming:
                if self.in_flight() >= self.depth * self.group:
                    break
                self.prefetch(nxt, flush=False)
            self._hold_launch = False
            y = y_all[pos * (x.shape[0] // 2):(pos + 1) * (x.shape[0] // 2)]
            return y if out is None else out.copy_(y)
        main = torch.cuda.current_stream()
        if not self._pending and self._waiting and self._waiting[0] is x:
            self._launch()                                   # end of a stream of batches: the group never filled
        if self._pending and isinstance(self._pending[0][0], list) and self._pending[0][0][0] is x:
            xs, payload, done = self._pending.popleft()
            pairs = x.shape[0] // 2
            want = len(xs) * pairs
            whole = out is not None and out.shape[0] == want
            if done is None:                                 # eager_dense: enqueued behind the group's sampling launch already
                y_all = payload
                if whole:
                    y_all = out.copy_(y_all)
            else:
                rows, prep = payload
                main.wait_event(done)
                for t in self._tensors((rows, prep)):
                    t.record_stream(main)
                with torch.no_grad():
                    y_all = self._model.merge_rows(rows, want, prep=prep, out=out if whole else None)
                if hasattr(prep, 'release'):
                    prep.release()                           # its buffers may be reused once these launches are through
            self._group_out = (list(xs[1:]), y_all)
            # the dense stages are enqueued (one foreign call); now the sampling group held back during the slice steps
            # and whatever else fits the pipeline depth
            if len(self._waiting) >= self.group:
                self._launch()
            for nxt in upcoming:
                if self.in_flight() >= self.depth * self.group:
                    break
                self.prefetch(nxt, flush=False)
            y = y_all[:pairs]
            return y if (out is None or whole) else out.copy_(y)
        dense_on = main
        if self._dense_streams:
            dense_on = self._dense_streams[self._dense_next]
            self._dense_next = (self._dense_next + 1) % len(self._dense_streams)
            if not self._inputs_ready or out is not None:
                dense_on.wait_stream(main)                   # whatever produced the batch / last used `out` on the caller's stream
        if self._pending and self._pending[0][0] is x:
            _, ready, done = self._pending.popleft()
            dense_on.wait_event(done)
            for t in self._tensors(ready):
                t.record_stream(dense_on)
        for nxt in upcoming:
            if self.in_flight() >= self.depth * self.group:
                break
            self.prefetch(nxt, flush=False)
        if ready is None:                                    # never sampled ahead: everything runs here, behind its producer
            ev = x.__dict__.pop('_dclr_ready', None)
            if ev is not None:
                dense_on.wait_event(ev)
        with torch.no_grad(), torch.cuda.stream(dense_on):
            prep = None
            if self._ahead == 'knn' and ready is not None:
                ready, prep = ready
            if ready is not None and self._ahead != 'sample':
                f_rows = ready
            else:
                f_rows = self._model.cloud_feature_rows(x, ready)
            y = self._dense(f_rows, x, prep, out)
        if dense_on is not main:
            finished = torch.cuda.Event()
            finished.record(dense_on)
            main.wait_event(finished)                        # the caller's stream sees the outputs in step order
            y.record_stream(main)
        return y

    @staticmethod
    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                yield from PipelinedForward._tensors(o)

    def _dense(self, f_rows: torch.Tensor, x: torch.Tensor, prep=None, out=None) -> torch.Tensor:
        y = self._model.merge_rows(f_rows, x.shape[0] // 2, prep=prep, out=out)
        if hasattr(prep, 'release'):
            prep.release()                                   # its buffers may be reused once these launches are through
        return y

    def run(self, batches: Iterable[torch.Tensor]) -> Iterator[torch.Tensor]:
        it = iter(batches)
        window: Deque[torch.Tensor] = deque()

        def refill():
            while len(window) < self.depth * self.group + 1:
                nxt = next(it, None)
                if nxt is None:
                    break
                window.append(nxt)

        refill()
        if not window:
            return
        self.prefetch(window[0])
        while window:
            started = self.in_flight()                       # counted while `cur` is still window[0]: window[:started] are
            cur = window.popleft()                           # the batches already handed to prefetch()
            refill()
            yield self.step(cur, list(window)[max(0, started - 1):])


class PipelinedSequence(PipelinedForward):
    """Odometry over one scan sequence, fed in chunks of T consecutive frames (T, N, C): every frame is
    sampled and abstracted ONCE and serves first as source, then as template of the next pair
    (the reference's sequential mode, /root/reference/deepclr/models/base.py:97-112, caches one frame and
    runs one pair per call). step() returns the poses frame[i-1] -> frame[i] for the chunk: (T, label_dim),
    or (T-1, label_dim) for the first chunk after reset().

    dense_group (with group > 1; round 6): the chunks sampled by one launch also share ONE dense launch -- the frames of
    the group are consecutive, so their pairs (plus the one across the border to the previous group, through the carried
    frame) go through layer-1 halves, kNN, flow embedding, head and fully connected tail together, enqueued when the
    group's first chunk is stepped; the other chunks of the group get slices of that result. A chunk of 16 frames alone is
    16 pairs per dense launch sequence (seven dependent launches on the caller's stream, which then set the pace, as in
    PipelinedForward without dense groups); ten chunks are 160."""

    def __init__(self, model: DeepCLR, depth: int = 3, ahead: str = 'features', group: int = 1, dense_group: bool = False):
        if ahead == 'knn':
            raise ValueError("pairs straddle chunk borders: the kNN stage cannot run per chunk ahead of time")
        super().__init__(model, depth, ahead, group)
        self._carry: Optional[torch.Tensor] = None
        self._seq_dense = bool(dense_group) and group > 1 and ahead == 'features'
        self._seq_out = None                        # (chunks of the running dense group not yet stepped, outputs, their spans)

    def reset(self) -> None:
        """Start a new sequence: no carried frame, and nothing left of a dense group whose chunks were not all stepped."""
        self._carry = None
        self._seq_out = None

    def in_flight(self) -> int:
        return super().in_flight() + (len(self._seq_out[0]) if self._seq_out is not None else 0)

    def _launch(self) -> None:
        if not self._seq_dense or len(self._waiting) < 2:
            return super()._launch()
        xs, self._waiting = self._waiting, []
        main = torch.cuda.current_stream()
        side = self._streams[self._next_stream]
        self._next_stream = (self._next_stream + 1) % self.depth
        if not self._inputs_ready:
            side.wait_stream(main)
        for b in xs:
            ev = b.__dict__.pop('_dclr_ready', None)
            if ev is not None:
                side.wait_event(ev)
        with torch.cuda.stream(side), torch.no_grad():
            big = torch.cat(xs)                                  # consecutive frames of the group's chunks
            rows = self._model.cloud_feature_rows(big, self._model.sample(big))
            done = torch.cuda.Event()
            done.record(side)
        for b in xs:
            b.record_stream(side)
        self._pending.append((xs, rows, done))

    def _refill(self, upcoming) -> None:
        for nxt in upcoming:
            if self.in_flight() >= self.depth * self.group:
                break
            self.prefetch(nxt, flush=False)

    def step(self, x: torch.Tensor, upcoming: Iterable[torch.Tensor] = (), out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self._seq_dense:
            return super().step(x, upcoming, out)
        if self._seq_out is not None and self._seq_out[0] and self._seq_out[0][0] is x:
            chunks, y_all, spans = self._seq_out                 # a chunk of the dense group already enqueued
            chunks.pop(0)
            lo, hi = spans.pop(0)
            self._hold_launch = True
            self._refill(upcoming)
            self._hold_launch = False
            y = y_all[lo:hi]
            return y if out is None else out.copy_(y)
        main = torch.cuda.current_stream()
        if not self._pending and self._waiting and self._waiting[0] is x:
            self._launch()                                       # end of the sequence: the group never filled
        if self._pending and isinstance(self._pending[0][0], list) and self._pending[0][0][0] is x:
            xs, rows, done = self._pending.popleft()
            main.wait_event(done)
            rows.record_stream(main)
            had_carry = self._carry is not None
            frames = sum(b.shape[0] for b in xs)
            with torch.no_grad():
                pair_rows, pairs, self._carry = self._model.sequence_rows(rows, frames, self._carry)
                y_all = self._model.merge_rows(pair_rows, pairs) if pairs > 0 else \
                    rows.new_empty(0, self._model.label_dim)
            spans, start = [], 0
            for j, b in enumerate(xs):                           # chunk j's poses: one per frame, minus the very first frame
                n = b.shape[0] - (0 if (had_carry or j > 0) else 1)
                spans.append((start, start + n))
                start += n
            self._seq_out = (list(xs[1:]), y_all, spans[1:])
            if len(self._waiting) >= self.group:
                self._launch()
            self._refill(upcoming)
            lo, hi = spans[0]
            y = y_all[lo:hi]
            return y if out is None else out.copy_(y)
        return super().step(x, upcoming, out)                    # a chunk sampled alone

    def _dense(self, f_rows: torch.Tensor, x: torch.Tensor, prep=None, out=None) -> torch.Tensor:
        pair_rows, pairs, self._carry = self._model.sequence_rows(f_rows, x.shape[0], self._carry)
        if pairs == 0:
            return f_rows.new_empty(0, self._model.label_dim)
        return self._model.merge_rows(pair_rows, pairs, out=out)


class HostBatchFeeder:
    """Batches that live in pinned HOST memory, copied to the device inside the loop on a copy stream of their own, so that
    the transfer of later batches overlaps the kernels of the current one -- what the reference does per pair with a
    blocking `.cuda()` (/root/reference/scripts/inference.py:89-90).

    Copies are made per CHUNK of `chunk` consecutive batches, (chunk, 2B, N, C) contiguous in pinned memory (a loader
    filling a pinned ring produces exactly that): on this stack a 4 MB pinned copy takes a slow path (9.9 GB/s and 0.34 ms
    of host time per call, scratch/h2d_probe.py) while 16 MB and more run at 53-55 GB/s on the DMA engines with 2-5 us per
    enqueue and no compute unit involved. A ring of device chunk buffers receives the copies; a buffer is rewritten only
    after the steps that consumed its batches have been enqueued (an event on the caller's stream, which by then has waited
    for the sampling launches that read it). The sampling launch of a batch waits for the event recorded behind its
    chunk's copy (PipelinedForward.prefetch(ready=...))."""

    def __init__(self, runner: PipelinedForward, example: torch.Tensor, chunk: Optional[int] = None,
                 slots: Optional[int] = None):
        self._runner = runner
        self.chunk = chunk if chunk is not None else max(runner.group, 4)
        ahead = runner.depth * runner.group + runner.group
        n = slots if slots is not None else -(-ahead // self.chunk) + 2
        shape = (self.chunk,) + tuple(example.shape)
        self._ring = [torch.empty(shape, dtype=example.dtype, device=example.device) for _ in range(n)]
        self._free_after = [None] * n               # event: every batch of the slot's previous contents consumed
        self._left = [0] * n                        # batches of the slot not yet stepped
        self._copy = torch.cuda.Stream()
        self._next = 0
        self._queue: Deque[Tuple[int, torch.Tensor]] = deque()     # copied (or copying), not yet stepped: (slot, batch view)
        self._offered = 0                           # how many of _queue the runner has taken into prefetch()
        self.bytes_copied = 0

    def room(self) -> bool:
        """True while another chunk can be uploaded (a ring slot whose batches have all been stepped is free)."""
        return self._left[self._next] == 0

    def feed(self, host_chunk: torch.Tensor) -> None:
        """Upload one pinned chunk (k <= chunk batches, (k, 2B, N, C) contiguous) and queue its batches."""
        if host_chunk.dim() != self._ring[0].dim() or host_chunk.shape[1:] != self._ring[0].shape[1:] \
                or host_chunk.shape[0] > self.chunk or not host_chunk.is_contiguous():
            raise RuntimeError("HostBatchFeeder.feed expects a contiguous (k <= {}, {}) chunk".format(
                self.chunk, 'x'.join(str(d) for d in self._ring[0].shape[1:])))
        slot = self._next
        if self._left[slot] != 0:
            raise RuntimeError("HostBatchFeeder: ring too small for the batches in flight")
        self._next = (self._next + 1) % len(self._ring)
        k = host_chunk.shape[0]
        dst = self._ring[slot][:k]
        if self._free_after[slot] is not None:
            self._copy.wait_event(self._free_after[slot])
        with torch.cuda.stream(self._copy):
            dst.copy_(host_chunk, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy)
        self.bytes_copied += host_chunk.numel() * host_chunk.element_size()
        self._left[slot] = k
        for j in range(k):
            view = dst[j]
            self._queue.append((slot, view))
            view._dclr_ready = ready                # picked up by whichever prefetch() (or step()) takes the batch

    def fill(self) -> None:
        """Start sampling for as many queued batches as the pipeline holds (call after the first feed()s)."""
        for _, view in list(self._queue)[self._offered:]:
            if self._runner.in_flight() >= self._runner.depth * self._runner.group:
                break
            self._runner.prefetch(view, flush=False)
            self._offered += 1

    def pending(self) -> int:
        return len(self._queue)

    def step(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Results of the oldest queued batch; queued batches the pipeline has room for start sampling first."""
        slot, cur = self._queue[0]
        before, cur_offered = self._runner.prefetched, self._offered > 0
        upcoming = [t for _, t in list(self._queue)[max(1, self._offered):]]
        y = self._runner.step(cur, upcoming=upcoming, out=out)
        self._offered += self._runner.prefetched - before
        self._queue.popleft()
        if cur_offered:
            self._offered -= 1
        self._left[slot] -= 1
        if self._left[slot] == 0:
            ev = torch.cuda.Event()
            ev.record()                              # caller's stream: it has waited for the launches that read the chunk
            self._free_after[slot] = ev
        return y
