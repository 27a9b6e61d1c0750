"""Multi-GPU plumbing of bench.py.

Default: ranks are independent replicas of the fused cycle batch — no data-path collective (SURVEY.md §8e "replicas");
torch.distributed only lines the ranks up (barrier) and takes the max elapsed time over ranks.

Optional: the LLM layer-sharded over the ranks (BASELINE config 5's partitioning): `LlmPipeline` gives rank r the
layers [r L / N, (r + 1) L / N) and moves the [rows, d_model] fp32 residual stream from stage to stage with point-to-point
send / recv — RCCL over the direct xGMI link between consecutive GPUs (backend "nccl"), or gloo through host memory in the
tests.  No all-reduce exists on this path.  Several row groups ("micro-batches") keep the stages busy: a sequence's next token
needs the previous one, so with fewer groups than stages the pipeline idles (§8e).

Backend "nccl" (= RCCL on ROCm) on the GPU box, "gloo" in the tests."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, local_rank=0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def barrier(dist, cuda):
    if cuda:
        import torch
        torch.cuda.synchronize()
    dist.barrier()
    if cuda:
        import torch
        torch.cuda.synchronize()


def max_over_ranks(dist, value, cuda):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cycle_seeds(rank, batch):
    """disjoint synthetic-input streams per rank: global cycle ids [rank*batch, (rank+1)*batch)"""
    return list(range(rank * batch, (rank + 1) * batch))


def aggregate_throughput(cycles_per_rank, steps, world, elapsed_max):
    """whole-job value: every rank processed cycles_per_rank * steps cycles in the (max) elapsed time"""
    return cycles_per_rank * steps * world / elapsed_max


class LlmPipeline:
    """Greedy generation with the layers of one model split over the ranks of `dist` (one process per GPU).

    Every rank builds the same model (same checkpoint / synthetic seed) and calls generate() with the same prompts; rank r only ever
    runs layers [bounds[r], bounds[r + 1]) of it.  Stage 0 embeds tokens, the last stage samples and sends the ids back to stage 0.
    All ranks walk the same (pass) order, sends are non-blocking, and stage 0 collects a group's ids only when it needs them for that
    group's next step — so with at least `world` row groups every stage always has a pass to work on.
    Results are bit-identical to a single-GPU LlmSession (the residual stream crosses GPUs as exact fp32)."""

    def __init__(self, dist, session, n_layer, d_model, cuda_tensors):
        import torch
        self.torch, self.dist, self.sess = torch, dist, session
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bounds = [(n_layer * r) // self.world for r in range(self.world + 1)]
        self.l0, self.l1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.first, self.last = self.rank == 0, self.rank == self.world - 1
        self.d_model = d_model
        self.cuda = cuda_tensors  # True: torch CUDA tensors are the RCCL send / recv buffers; False: host tensors (gloo)
        self._inflight = []       # (work, tensor) of non-blocking sends: the tensor must outlive the transfer

    def _buf(self, n):
        return self.torch.empty((n, self.d_model), dtype=self.torch.float32, device="cuda" if self.cuda else "cpu")

    def _ptr(self, t):
        return t.data_ptr() if self.cuda else t.numpy()

    def _isend(self, t, dst):
        self._inflight = [(w, x) for (w, x) in self._inflight if not w.is_completed()]
        self._inflight.append((self.dist.isend(t, dst=dst), t))

    def _drain(self):
        for w, _ in self._inflight:
            w.wait()
        self._inflight = []

    def _stage(self, seq, pos, tok, head):
        """this rank's layers for one pass: receive the stream (not stage 0), run, hand it on (not the last stage); ids when sampling"""
        n = len(seq)
        x_in = None
        if not self.first:
            x_in = self._buf(n)
            self.dist.recv(x_in, src=self.rank - 1)
            if self.cuda:
                self.torch.cuda.current_stream().synchronize()
        sample = head and self.last
        x_out = None if sample else self._buf(n)  # a prompt pass on the last stage still writes (and drops) its stream
        am = self.sess.forward_stage(seq, pos, self.l0, self.l1, tok=tok if self.first else None, x_in=None if x_in is None else self._ptr(x_in),
                                     x_out=None if x_out is None else self._ptr(x_out), head=sample, on_host=not self.cuda)
        if not self.last:
            self._isend(x_out, self.rank + 1)
        return am

    def _send_ids(self, ids):
        if self.world > 1:
            t = self.torch.from_numpy(ids.copy())
            self._isend(t.cuda() if self.cuda else t, 0)

    def _post_recv_ids(self, n):
        """stage 0: post the receive of a group's sampled ids as soon as that group's pass has left this stage — a send that finds its
        receive posted does not depend on the transport buffering it eagerly (any number of row groups may be in flight)"""
        t = self.torch.empty(n, dtype=self.torch.int32, device="cuda" if self.cuda else "cpu")
        return (self.dist.irecv(t, src=self.world - 1), t)

    def _wait_ids(self, posted):
        work, t = posted
        work.wait()
        if self.cuda:
            self.torch.cuda.current_stream().synchronize()
        return t.cpu().numpy()

    def generate(self, prompts, n_steps, rows_per_pass=128):
        """prompts: list of int32 arrays [nseq_g, n_prompt], one per row group (micro-batch; the sequences of group g use the cache
        slots after those of the groups before it).  Returns a list of [n_steps, nseq_g] arrays — the greedy continuations, the first
        token sampled from the last prompt token — filled in on rank 0 and on the last rank (zeros elsewhere)."""
        import numpy as np
        groups = [np.ascontiguousarray(p, np.int32) for p in prompts]
        base = np.cumsum([0] + [g.shape[0] for g in groups])
        out = [np.zeros((n_steps, g.shape[0]), np.int32) for g in groups]
        cur = [None] * len(groups)      # the ids a group feeds next (rank 0 and last rank)
        owed = [None] * len(groups)     # rank 0: the posted receive of this group's ids, on their way from the last stage
        for gi, g in enumerate(groups):
            nseq, n_prompt = g.shape
            seq = np.repeat(np.arange(nseq, dtype=np.int32) + base[gi], n_prompt - 1)
            pos = np.tile(np.arange(n_prompt - 1, dtype=np.int32), nseq)
            tok = g[:, :-1].reshape(-1)
            for i in range(0, len(seq), rows_per_pass):  # prompt rows whose logits nobody reads: every stage appends its K / V
                self._stage(seq[i:i + rows_per_pass], pos[i:i + rows_per_pass], tok[i:i + rows_per_pass], head=False)
            ids = self._stage(np.arange(nseq, dtype=np.int32) + base[gi], np.full(nseq, n_prompt - 1, np.int32), g[:, -1], head=True)
            if self.last:
                cur[gi] = ids
                if not self.first:
                    self._send_ids(ids)
            elif self.first:
                owed[gi] = self._post_recv_ids(nseq)
        for step in range(n_steps):
            for gi, g in enumerate(groups):
                nseq, n_prompt = g.shape
                if self.first and owed[gi] is not None:
                    cur[gi] = self._wait_ids(owed[gi])
                    owed[gi] = None
                if cur[gi] is not None:
                    out[gi][step] = cur[gi]
                if step + 1 == n_steps:
                    continue
                ids = self._stage(np.arange(nseq, dtype=np.int32) + base[gi], np.full(nseq, n_prompt + step, np.int32), cur[gi], head=True)
                if self.last:
                    cur[gi] = ids
                    if not self.first:
                        self._send_ids(ids)
                elif self.first:
                    owed[gi] = self._post_recv_ids(nseq)
        self._drain()
        return out
