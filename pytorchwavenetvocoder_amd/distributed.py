# -*- coding: utf-8 -*-
"""Data-parallel gradient reduction: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (train.py:449-454: per-step
parameter broadcast, logits gather to GPU0, gradient reduce to GPU0) with the minimum exchange the
path needs: every rank computes its own loss on its own shard of the minibatch and ONE flat fp32
gradient buffer is summed across ranks (gradients are pre-scaled by 1/world_size inside the loss
kernel, so the sum is the global-batch mean the reference's loss produces).  The buffer is laid
out in backward-completion order, so buckets are contiguous ranges: ``wn_backward`` records a HIP
event after each bucket and the all-reduce of bucket i runs on a side stream while the backward
kernels of the following layers are still executing.
"""
import os

import torch
import torch.distributed as dist


def rccl_footprint_defaults():
    """Call before ``init_process_group("nccl")``.  The fused chain kernels are persistent grids that need a whole CU
    per workgroup (120 KB LDS, every VGPR); for the benchmark geometry they use 240 of the 256 CUs.  An RCCL kernel
    with more workgroups (= channels) than the 16 CUs left would take CUs a chain kernel is about to claim, and the
    unplaced part of that grid only starts when the rest of it retires (measured with a side-stream contraction:
    DESIGN.md 5.1).  The gradient buffer is 6.4 MB in 4 buckets -- latency-bound, so 16 channels cost nothing.
    ``setdefault``: an explicit NCCL_MAX_NCHANNELS in the environment wins."""
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")


class GradientReducer(object):
    def __init__(self, model, process_group=None, layers_per_bucket=None, exchange_when_alone=False):
        """``layers_per_bucket``: residual layers per gradient bucket = per weight-gradient launch group of wn_backward.
        Default (None): chosen by the size of the gradient.
          * small models (the BASELINE 64 / 256 model: 6.4 MB): ALL layers in one bucket, i.e. three buckets
            [post-net] [skip_1x1 + residual layers] [front + upsampling] (wn_bucket_range: with one layer bucket the skip tensors
            belong to it -- their gradients are made behind the chain, in the launch that also makes the res_1x1 gradients).
            Measured on MI355X at N = 1
            (profiles/r03/visit3_chain_dw_pipelined_lpb_chainpairdiff.txt): 10.42 ms per step with one layer bucket, 10.49 with
            two (15 layers each), 10.64 with three -- the weight-gradient contractions are most efficient as ONE layer-batched
            launch per tensor kind.  What a split would hide is the all-reduce of the 5.7 MB layer bucket, a latency-bound
            ring of ~0.1 ms over xGMI: splitting costs as much compute as it could hide, so it is not done; the post-net
            bucket travels under the whole backward chain either way.
          * large models (gradient above 32 MB; the recipe-size 512 / 256 model: 185 MB, ~2 - 3 ms of ring time against a
            131 ms step): groups of 10 layers, so that two thirds of the layer gradients travel under the rest of the chain;
            at that width the launch-group cost is below 0.5 % of the step."""
        self.model = model
        self.group = process_group
        self.eng = model.engine
        if layers_per_bucket:
            self.lpb = int(layers_per_bucket)
        else:
            big = 4 * int(self.eng.n_params) > 32 * 1024 * 1024
            self.lpb = min(10, int(self.eng.n_layers)) if big else int(self.eng.n_layers)
        self.ranges = self.eng.bucket_ranges(self.lpb)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # tests: run the bucket-event / side-stream / all-reduce branch with a single rank too (RCCL then executes, on this GPU)
        self.exchange_alone = bool(exchange_when_alone) and dist.is_initialized()
        self.cuda = self.eng.device.type == "cuda"
        if self.cuda:
            self.side = torch.cuda.Stream(device=self.eng.device)
            self.events = [torch.cuda.Event() for _ in self.ranges]
            for e in self.events:  # create the underlying hipEvent_t handles
                e.record(torch.cuda.current_stream(self.eng.device))
        else:
            self.side, self.events = None, None
        # opt-in measurement of the EXPOSED part of the exchange (bench.py's `comm` block): an event after the last backward
        # launch and one after the caller's stream has joined the side stream; their distance is what the all-reduces did NOT
        # hide under the backward kernels
        self.measure_exposed = False
        self._exposed = []

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def _step(self, x, h, t, y, **kw):
        if y is not None:   # mixture-of-logistics head: the target is the waveform value, not the token
            return self.model.mol_loss_and_backward(x, h, y, **kw)
        return self.model.loss_and_backward(x, h, t, **kw)

    def loss_and_backward(self, x, h, t, t_start=None, y=None, grad_scale=None):
        """forward + loss + backward with the bucketed all-reduce overlapped; returns the local
        mean loss (device tensor).  ``y`` (B, T) float selects the mixture-of-logistics loss.
        ``grad_scale``: this rank's share of the global minibatch (default 1/world = equal shards; pass
        B_local / B_global when the shards are uneven, so that the summed gradient is the global-batch mean)."""
        if self.world == 1 and not self.exchange_alone:   # same launch structure as N > 1 (weight gradients flushed per bucket), no exchange
            return self._step(x, h, t, y, t_start=t_start, layers_per_bucket=self.lpb)
        gscale = self.grad_scale if grad_scale is None else float(grad_scale)
        if not self.cuda:
            loss = self._step(x, h, t, y, t_start=t_start, grad_scale=gscale)
            flat = self.eng.grads()
            for lo, hi in self.ranges:
                dist.all_reduce(flat[lo:hi], group=self.group)
            return loss
        handles = [e.cuda_event for e in self.events]
        loss = self._step(x, h, t, y, t_start=t_start, grad_scale=gscale, events=handles,
                          layers_per_bucket=self.lpb)
        flat = self.eng.grads()
        main = torch.cuda.current_stream(self.eng.device)
        pair = None
        if self.measure_exposed:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record(main)     # behind the last backward launch
        with torch.cuda.stream(self.side):
            for (lo, hi), ev in zip(self.ranges, self.events):
                self.side.wait_event(ev)
                dist.all_reduce(flat[lo:hi], group=self.group)
        main.wait_stream(self.side)
        if pair is not None:
            pair[1].record(main)     # the caller's stream has joined the exchange
            self._exposed.append(pair)
        return loss

    def comm_report(self):
        """What the exchange looked like (after a synchronize): ranks and backend as torch.distributed reports them, bucket
        sizes, the RCCL channel cap in effect, and -- over the steps taken with ``measure_exposed`` -- the time the caller's
        stream waited for the side stream after its last backward launch (exposed exchange per step, ms)."""
        rep = {"world": self.world, "backend": dist.get_backend(self.group) if dist.is_initialized() else None,
               "buckets": len(self.ranges), "bucket_bytes": [4 * int(hi - lo) for lo, hi in self.ranges],
               "layers_per_bucket": self.lpb, "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
               "exchange": "one in-place all-reduce per bucket on a side stream, released by the bucket's HIP event"
                           if (self.world > 1 or self.exchange_alone) else "none (one rank)"}
        if self._exposed:
            ms = [a.elapsed_time(b) for a, b in self._exposed]
            rep["exposed_ms_per_step"] = {"mean": sum(ms) / len(ms), "min": min(ms), "max": max(ms), "steps": len(ms)}
            self._exposed = []
        return rep
