"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient buffer, overlapped with backward.

Reference: torch DistributedDataParallel wrapped around the model (detectron2/engine/defaults.py:279-285; NCCL).
Here: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests); the
gradients already live in one contiguous buffer in backward order (solver.FlatSGD), cut into a few large buckets
(default 32 MB: one node's xGMI mesh is point-to-point, large messages keep every link busy).  A bucket's
all-reduce is launched from the autograd hook of the last parameter that becomes ready in it, so communication
runs while earlier layers are still back-propagating.  Gradients are pre-divided by the world size on the producer
side (ops_train.PARAM_GRAD_DIV), so SUM gives the mean like DDP; BatchNorm statistics and loss normalisers stay
per-GPU like the reference.
"""
import torch
import torch.distributed as dist


class BucketedReducer:
    def __init__(self, optimizer, bucket_bytes=32 << 20, process_group=None, comm=None):
        """comm: an engine.rccl.RcclComm -- the buckets then go through the C ABI (`ctdet_allreduce_bucket`) on a side
        stream instead of torch.distributed's collectives (the process group, if any, only bootstrapped the communicator)"""
        self.opt, self.group, self.comm = optimizer, process_group, comm
        self._side = None
        if comm is not None:
            self.world = comm.world            # the communicator that carries the exchange says how many ranks there are
        else:
            self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = []  # [start, end, n_params]
        self.bucket_of = []
        cur = None
        for i, (off, n) in enumerate(optimizer.offsets):
            if cur is None or (off + n - cur[0]) * 4 > bucket_bytes and cur[2] > 0:
                cur = [off, off, 0]
                self.buckets.append(cur)
            cur[1] = off + n
            cur[2] += 1
            self.bucket_of.append(len(self.buckets) - 1)
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self.calls = {}       # parameters announced in the current backward pass
        self.enabled = True   # False while a training step is captured / replayed as a graph: see reduce_all()
        if self.world > 1:
            for i, p in enumerate(optimizer.params):
                h = self._make_hook(i)
                p.register_post_accumulate_grad_hook(h)
                p._ctdet_grad_hook = h    # backward kernels that accumulate into the flat buffer themselves announce it here

    def _make_hook(self, i):
        b = self.bucket_of[i]

        def hook(param):
            if not self.enabled:
                return
            # a parameter is announced by autograd's AccumulateGrad, by the backward wrapper that wrote its gradient into the
            # flat buffer itself (ops_train.grad_done), or by both (autograd also visits the node when the function returned
            # None): the first announcement of a pass counts
            if i in self.calls:
                return
            self.calls[i] = 1
            self._ready[b] += 1
            if self._ready[b] == self.buckets[b][2] and not self._launched[b]:
                self._launch(b)
        hook.reducer = self       # lets a caller that runs a backward pass outside the trainer switch the exchange off
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        if self.opt.flat_grad.is_cuda:      # weight gradients still waiting in tap-major form for this range (ops_train.PENDING)
            from .. import ops_train
            base = self.opt.flat_grad.data_ptr()
            ops_train.SIDE.join()           # weight gradients still in flight on the side stream
            ops_train.flush_param_grads(base + 4 * s, base + 4 * e)
        if self.comm is not None:
            # RCCL through the C ABI: ordered after the gradients produced so far, on a side stream so that the rest of
            # backward keeps running; finish() joins the streams
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            self.comm.all_reduce_(self.opt.flat_grad[s:e], stream=self._side)
            return
        self._handles.append(dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))

    def prepare(self):
        """call before backward"""
        self.calls = {}
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []

    def finish(self):
        """call after backward: launches buckets that contain unused parameters, waits for all of them"""
        if self.world == 1:
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.comm is not None and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def reduce_all(self):
        """all buckets, back to back, after a backward pass that ran without the hooks (the captured-graph step: forward +
        backward replay as one HIP graph, the exchange and the SGD launch follow it; the hooks only exist while autograd
        runs eagerly)"""
        if self.world == 1:
            return
        if self.comm is not None:
            for s, e, _ in self.buckets:
                self.comm.all_reduce_(self.opt.flat_grad[s:e])
            return
        handles = [dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for s, e, _ in self.buckets]
        for h in handles:
            h.wait()

    def broadcast_parameters(self, buffers=()):
        """DDP's initial synchronisation: rank 0's parameters (and BN buffers) to everyone."""
        if self.world == 1:
            return
        if self.comm is not None:
            self.comm.broadcast_(self.opt.flat_param, 0)
            for b in buffers:
                if b.dtype == torch.float32 and b.is_contiguous():
                    self.comm.broadcast_(b.view(-1), 0)
                else:
                    dist.broadcast(b, src=0, group=self.group)
            return
        dist.broadcast(self.opt.flat_param, src=0, group=self.group)
        for b in buffers:
            dist.broadcast(b, src=0, group=self.group)
