"""SimpleTrainer.run_step (detectron2/engine/train_loop.py:212-251) for the HIP training path."""
import os
import time

import torch

from .. import ops_train
from ..solver import WarmupMultiStepLR, build_optimizer
from . import graph_nodes, train_step
from .reducer import BucketedReducer


MAX_TRAIN_GRAPHS = int(os.environ.get("CTDET_MAX_TRAIN_GRAPHS", "4"))


class SimpleTrainer:
    """model(data) -> loss dict; sum; zero_grad; backward (bucketed all-reduce overlapped); SGD step; LR schedule.

    Per-iteration metrics are kept on the device and only fetched when `metrics()` is called: the reference does a
    `.item()` sync plus a Gloo pickle gather every step (train_loop.py:261-290), which this build avoids."""

    def __init__(self, model, data_loader, cfg, optimizer=None, process_group=None):
        model.train()
        self.model, self.cfg = model, cfg
        self._data_iter = iter(data_loader) if data_loader is not None else None
        self.optimizer = optimizer or build_optimizer(cfg, model)
        self.reducer = BucketedReducer(self.optimizer, process_group=process_group, comm=self._rccl_comm(process_group))
        ops_train.set_world_size(self.reducer.world)
        self.reducer.broadcast_parameters([b for b in model.buffers() if b.dtype.is_floating_point])
        self.scheduler = WarmupMultiStepLR(self.optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR,
                                           cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_METHOD)
        self.iter = 0
        self.last_losses = None
        import os
        # CTDET_TRAIN_GRAPH: "1" (default): the step replays as a HIP graph -- single GPU: targets, forward, losses, backward and
        # the SGD launch in one graph; data parallel: forward + backward in the graph, then the bucketed all-reduce, the SGD launch
        # and the batched weight re-pack behind it.  "hooks": data-parallel steps stay eager with the all-reduce of a bucket
        # launched from autograd hooks as soon as its gradients are complete (overlaps backward; ~1,000 launches per step and
        # rank).  "0": every step eager.  ("ddp", round 3's name of the data-parallel graph path, is accepted as "1".)
        mode = os.environ.get("CTDET_TRAIN_GRAPH", "1")
        self.use_hip_graph = mode != "0"
        self.graph_ddp = mode not in ("0", "hooks")
        self._graphs = {}

    @staticmethod
    def _rccl_comm(process_group):
        """CTDET_RCCL_DIRECT=1: the gradient exchange goes through the C ABI's RCCL entry points (ctdet_comm_init /
        ctdet_allreduce_bucket, engine/rccl.py) instead of torch.distributed's collectives; the process group only carries the
        128-byte unique id.  One rank per GPU (RCCL cannot put two ranks on one device)."""
        import os
        import torch.distributed as dist
        if os.environ.get("CTDET_RCCL_DIRECT", "0") != "1" or not torch.cuda.is_available():
            return None
        if not (dist.is_available() and dist.is_initialized()):
            from .rccl import RcclComm
            return RcclComm(0, 1)
        from .rccl import RcclComm
        return RcclComm(dist.get_rank(process_group), dist.get_world_size(process_group), process_group)

    def run_step(self, data=None):
        assert self.model.training, "[SimpleTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        if data is None:
            data = next(self._data_iter)
        self.data_time = time.perf_counter() - start
        return self._finish_step(self.model(data))

    def run_step_tensors(self, images, boxes, classes, counts):
        """device-resident batch (uint8 [B,3,H,W], boxes f32 [B,N,4], classes i64 [B,N], counts i32 [B]).

        From the third call with a given batch shape on, the step replays as ONE captured HIP graph: the eager step issues
        ~1,000 launches (23.9 ms against 20.5 replayed on an idle host; eight ranks' launch threads share one host).
        Single GPU: targets, forward, losses, backward and the SGD launch are all in the graph.  Data parallel: forward +
        backward replay as a graph, the all-reduce of the flat gradient buffer (three 32 MB buckets, back to back), the SGD
        launch and the weight re-pack follow it (CTDET_TRAIN_GRAPH=hooks: eager, buckets launched from autograd hooks so that
        they overlap backward).  Two ranks on one device == the single-process trajectory to 1e-7
        (tests/test_dp_gpu.py::test_two_rank_graph_steps_equal_single_rank_steps).  The LR schedule and the per-parameter
        version counters stay on the host.  CTDET_TRAIN_GRAPH=0 keeps every step eager."""
        multi = self.reducer.world > 1
        if not self.use_hip_graph or (multi and not self.graph_ddp):
            return self._finish_step(self.model.train_batch_tensor(images, boxes, classes, counts))
        key = tuple((tuple(t.shape), t.dtype) for t in (images, boxes, classes, counts))
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = {"calls": 0, "graph": None}
        else:
            self._graphs[key] = self._graphs.pop(key)          # most recently used last
        g["calls"] += 1
        if g["graph"] is None and g["calls"] <= 2 or g.get("failed"):
            return self._finish_step(self.model.train_batch_tensor(images, boxes, classes, counts))
        if g["graph"] is None:
            # a captured step owns a private memory pool the size of the step's working set (gigabytes): multi-scale training
            # (MIN_SIZE_TRAIN with six sizes, ragged aspect ratios) would pile them up.  At most MAX_TRAIN_GRAPHS stay; the
            # least recently used one is destroyed here, outside any capture, and its shape goes back to eager steps until
            # it has been seen twice again.  Entries of shapes that never reached a capture are only counters.
            live = [k for k, e in self._graphs.items() if e["graph"] is not None]
            while len(live) >= MAX_TRAIN_GRAPHS:
                old = self._graphs.pop(live.pop(0))
                old["graph"] = old["inputs"] = old["losses"] = None
                del old
            if len(self._graphs) > 64:                          # counters of shapes seen once or twice
                for k in [k for k, e in self._graphs.items() if e["graph"] is None and k != key][:len(self._graphs) - 64]:
                    del self._graphs[k]
            self._capture(g, images, boxes, classes, counts, with_step=not multi)
            if g.get("failed"):
                return self._finish_step(self.model.train_batch_tensor(images, boxes, classes, counts))
        for dst, src in zip(g["inputs"], (images, boxes, classes, counts)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        g["graph"].replay()
        if train_step.STEP_CHECK == 1:
            self._check_step("after the replayed forward + backward" + ("" if multi else " + SGD"), g["losses"])
        elif train_step.STEP_CHECK == 2:
            self._log_step(7, g["losses"])
        if multi:
            self.reducer.reduce_all()
            if train_step.STEP_CHECK == 1:
                self._check_step("after the all-reduce", g["losses"])
            elif train_step.STEP_CHECK == 2:
                self._log_step(8)
            self.optimizer.step()
        if train_step.STEP_CHECK == 2:
            self._log_step(9)
        for p in self.optimizer.params:   # the captured SGD kernels wrote the parameters behind autograd's back
            torch.autograd.graph.increment_version(p)
        self.scheduler.step()
        self.iter += 1
        self.last_losses = g["losses"]
        return self.last_losses

    def _log_step(self, col, losses=None):
        """CTDET_TRAIN_CHECK=2: one row per step in a device tensor, written by stream-ordered launches only (no host sync, so
        the timing of the step stays what it is): [hm, wh, off, max|features|, max|logits|, max target, min target,
        gradients finite after backward, after the all-reduce, parameters finite after SGD]; `step_log()` reads it"""
        if getattr(self, "_steplog", None) is None:
            self._steplog = torch.full((4096, 10), -1.0, device=self.optimizer.flat_param.device)
        row = self._steplog[self.iter % 4096]
        if losses is not None:
            row[0:3].copy_(torch.stack([losses[k].float().reshape(()) for k in ("hm_loss", "wh_loss", "off_loss")]))
            st = train_step.STEP_STATS.get("forward")
            if st is not None:
                row[3:7].copy_(st)
        src = self.optimizer.flat_param if col == 9 else self.optimizer.flat_grad
        row[col:col + 1].copy_(torch.isfinite(src).all().float().reshape(1))

    def step_log(self):
        torch.cuda.synchronize()
        return self._steplog[:self.iter].cpu()

    def _check_step(self, where, losses):
        """CTDET_TRAIN_CHECK=1 (debug; one host sync per check): the first step whose losses, forward statistics or flat
        gradient / parameter buffers hold a non-finite value raises, naming the rank's step, the place and the parameters"""
        torch.cuda.synchronize()
        opt = self.optimizer
        loss = {k: float(v) for k, v in losses.items()}
        stats = train_step.STEP_STATS.get("forward")
        stats = [float(v) for v in stats] if stats is not None else []
        ok_g = bool(torch.isfinite(opt.flat_grad).all())
        ok_p = bool(torch.isfinite(opt.flat_param).all())
        fin = all(v == v and abs(v) != float("inf") for v in list(loss.values()) + stats)
        if fin and ok_g and ok_p:
            return
        names = {id(p): n for n, p in self.model.named_parameters()}
        bad_g = [names.get(id(p), "?") for p in opt.params if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        bad_p = [names.get(id(p), "?") for p in opt.params if not bool(torch.isfinite(p).all())]
        raise FloatingPointError(
            f"[rank {os.environ.get('RANK', '0')}] step {self.iter} {where}: losses {loss}; "
            f"max|features|, max|logits|, max target, min target = {stats}; {len(bad_g)} non-finite gradients "
            f"(first {bad_g[:6]}, last {bad_g[-3:]}); {len(bad_p)} non-finite parameters (first {bad_p[:6]})")

    def _capture(self, g, images, boxes, classes, counts, with_step=True):
        import gc
        inputs = [t.clone() for t in (images, boxes, classes, counts)]
        torch.cuda.synchronize()
        gc.collect()
        gc_on = gc.isenabled()
        gc.disable()
        self.reducer.enabled = False     # no collective may be launched from a hook while the stream is capturing
        import warnings
        try:
            graph = torch.cuda.CUDAGraph(keep_graph=True)     # instantiated in _capture_body, after a look at its nodes
            # An AccumulateGrad node that an older autograd graph keeps alive runs on the stream IT was created on (the default
            # stream) and would fork the capture onto that stream: hipStreamEndCapture crashes on such a capture (round 3's
            # VoVNet segfault; the step's own nodes write their gradients straight into the flat buffer and need no
            # AccumulateGrad).  torch announces the situation with a warning: inside the capture it is an error, the capture is
            # abandoned before the fork and the step stays eager (graph_state "failed", traceback logged).
            with warnings.catch_warnings():
                warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
                self._capture_body(graph, inputs, g, with_step)
        except RuntimeError as e:
            self._capture_failed(g, e)
        except Warning as e:
            self._capture_failed(g, RuntimeError(f"stale autograd graph alive during capture: {e}"))
        finally:
            self.reducer.enabled = True
            if gc_on:
                gc.enable()

    def _capture_failed(self, g, e):
        # capture is an optimisation, but a failed one must be visible: the step falls back to eager launches, the state
        # is reported by `graph_state` (bench.py prints it) and the traceback is logged once
        import logging
        import traceback
        g["failed"] = repr(e)
        g["traceback"] = traceback.format_exc()
        logging.getLogger(__name__).warning("HIP-graph capture of the training step failed; running eagerly.\n%s",
                                            g["traceback"])
        torch.cuda.synchronize()

    def _capture_body(self, graph, inputs, g, with_step):
        with torch.cuda.graph(graph):
            loss_dict = self.model.train_batch_tensor(*inputs)
            losses = sum(loss_dict.values())
            self.optimizer.zero_grad()
            losses.backward()
            if with_step:
                self.optimizer.step()
        # The captured step must be kernels only.  A hipMemsetAsync / device-to-device hipMemcpyAsync issued inside it becomes a
        # memset / memcpy node, and a replay launched on an idle stream ran such a node out of order with the kernel after it
        # (the data-parallel step, whose SGD launch is eager: round 3's "CTDET_TRAIN_GRAPH=ddp gives inf hm_loss"; engine/
        # graph_nodes.py).  Anything else found here is reported, loudly, once per capture.
        g["nodes"] = graph_nodes.inspect(graph, "training step") or {}
        graph.instantiate()
        g["graph"], g["inputs"] = graph, inputs
        g["losses"] = {k: v.detach() for k, v in loss_dict.items()}
        # the capture itself executed nothing: this call's step is the first replay

    @property
    def graph_state(self):
        """"captured": the last tensor step replayed a HIP graph; "failed": a capture raised (eager fallback, traceback
        logged and kept in _graphs[key]["traceback"]); "eager": no capture attempted (yet)"""
        states = [("failed" if g.get("failed") else "captured" if g["graph"] is not None else "eager")
                  for g in self._graphs.values()]
        if "failed" in states:
            return "failed"
        return "captured" if "captured" in states else "eager"

    def _finish_step(self, loss_dict):
        losses = sum(loss_dict.values())
        self.optimizer.zero_grad()
        self.reducer.prepare()
        losses.backward()
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        self.last_losses = {k: v.detach() for k, v in loss_dict.items()}
        if train_step.STEP_CHECK == 1:
            self._check_step("after the eager step", self.last_losses)
        return self.last_losses

    def metrics(self):
        """host copy of the last losses; raises like train_loop.py:253-259 on non-finite values"""
        out = {k: float(v) for k, v in self.last_losses.items()}
        if not all(torch.isfinite(torch.tensor(list(out.values())))):
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {out}")
        return out
