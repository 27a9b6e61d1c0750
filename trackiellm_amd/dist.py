"""Multi-GPU plumbing of bench.py.

Default: ranks are independent replicas of the fused cycle batch — no data-path collective (SURVEY.md §8e "replicas");
torch.distributed only lines the ranks up (barrier) and takes the max elapsed time over ranks.

Optional: the LLM layer-sharded over the ranks (BASELINE config 5's partitioning): `LibPipeline` gives rank r the
layers [r L / N, (r + 1) L / N); the [rows, d_model] residual stream moves from stage to stage INSIDE the library (device mailboxes
mapped with hipIpc, csrc/llm/tk_llm_pipe.h) — torch.distributed only carries the 80-byte mailbox handles once.  No all-reduce exists on
this path.  Several row groups ("micro-batches") keep the stages busy: a sequence's next token needs the previous one, so with fewer
groups than stages the pipeline idles (§8e).

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


MAX_DETS = 20  # detections per frame that reach the prompt (the reference's max_detected_objects, src/cortex/tk_cortex_main.c:781)


def pack_perception(dets, tokens):
    """what a perception rank hands to the LLM's first rank for its share of a cycle batch — the bytes the reference turns into the
    prompt's context (src/cortex/tk_cortex_main.c:1224-1237, 1323-1345): per frame up to MAX_DETS (class, confidence, x, y, w, h) rows,
    per utterance the decoded token ids.  dets: list (per frame) of detection lists / tuples, or None; tokens: int array [n][steps] or None."""
    import numpy as np
    out = {}
    if dets is not None:
        a = np.zeros((len(dets), MAX_DETS, 6), np.float32)
        n = np.zeros(len(dets), np.int32)
        for i, fr in enumerate(dets):
            k = min(len(fr), MAX_DETS)
            n[i] = k
            for j in range(k):
                d = fr[j]
                # vision.ObjectDetector.detect_batch: (class id, label, confidence, (x, y, w, h)); or six plain numbers
                a[i, j] = (d[0], d[2], *d[3]) if len(d) == 4 and isinstance(d[3], (tuple, list)) else tuple(d)[:6]
        out["dets"], out["n_dets"] = a, n
    if tokens is not None:
        out["tokens"] = np.ascontiguousarray(tokens, np.int32)
    return out


class PerceptionExchange:
    """The cycle's data dependency across ranks (model-per-gpu and combined placements): every step ends with ONE gather of the
    perception ranks' results of the batch they just finished to the LLM's first rank (rank `dst`), which must hold them before it
    generates for that batch — the software pipeline of the one-GPU bench (LLM of batch k beside the perception of batch k + 1) with its
    hand-off made explicit.  Every rank of the job calls hand_over() once per step, in the same place; the payload is a few hundred
    bytes per cycle and goes through torch.distributed (gather_object: any backend)."""

    def __init__(self, dist, dst=0):
        self.dist, self.dst = dist, dst
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.received = None      # on dst: what the perception ranks produced last step
        self.bytes_last = 0
        self.checksum = 0

    def hand_over(self, payload):
        """payload: pack_perception(...) on perception ranks, None elsewhere.  On `dst` the gathered results become `received`."""
        import numpy as np
        box = [None] * self.world if self.rank == self.dst else None
        self.dist.gather_object(payload, box, dst=self.dst)
        if self.rank != self.dst:
            return None
        got = [b for b in box if b]
        self.received = got
        self.bytes_last = int(sum(v.nbytes for b in got for v in b.values()))
        cs = 0
        for b in got:  # consumed: every value is read once (the prompt builder's stand-in; the token ids of the prompt stay the fixed
            for v in b.values():  # seeded ones of SURVEY.md 8d, so the numbers compare with the one-GPU run)
                cs = (cs * 1000003 + int(np.asarray(v, np.float64).sum() * 16)) % (1 << 61)
        self.checksum = cs
        return got

    def require(self, n_frames, n_utts):
        """on dst, before generating for a batch: its perception results are here and complete"""
        got = self.received or []
        f = sum(len(b["n_dets"]) for b in got if "n_dets" in b)
        u = sum(len(b["tokens"]) for b in got if "tokens" in b)
        if f < n_frames or u < n_utts:
            raise RuntimeError("perception results of this batch are missing: %d / %d frames, %d / %d utterances" % (f, n_frames, u, n_utts))


def combined_roles(world):
    """who does what in a job of `world` ranks (one per GPU): {"llm": the pipeline's ranks in stage order, "vision": detector ranks,
    "audio": VAD + ASR ranks}.
      1: everything on rank 0            2: LLM on 0, both perception streams on 1      3: two LLM stages, perception on 2
      4 and more: ranks 0 .. world - 3 form the pipeline, rank world - 2 runs the detector, rank world - 1 VAD + ASR
      (8: six stages of 5 - 6 layers + detector + ASR — the row SURVEY.md §8e names)"""
    if world <= 1:
        return {"llm": [0], "vision": [0], "audio": [0]}
    if world == 2:
        return {"llm": [0], "vision": [1], "audio": [1]}
    if world == 3:
        return {"llm": [0, 1], "vision": [2], "audio": [2]}
    return {"llm": list(range(world - 2)), "vision": [world - 2], "audio": [world - 1]}


def stage_bounds(n_layer, n_stages):
    """layers [bounds[s], bounds[s + 1]) belong to stage s: as even as integer division allows, every stage at least one layer"""
    if not 1 <= n_stages <= n_layer:
        raise ValueError("a pipeline needs between 1 and n_layer stages")
    return [(n_layer * s) // n_stages for s in range(n_stages + 1)]


def stage_schedule(prompts, n_steps, rows_per_pass):
    """the pass list EVERY stage enqueues, in this order, for its row groups: per group the prompt chunks (no sampling), the sampling pass
    and the decode loop.  prompts: list of int arrays [nseq_g][n_prompt_g].  Entries: ("pass", g, seq, pos, tok, head) / ("decode", g, nseq, n_steps).
    Sequences are numbered inside their group (each group has a session of its own on every stage)."""
    import numpy as np
    out = []
    for gi, g in enumerate(prompts):
        g = np.ascontiguousarray(g, np.int32)
        nseq, n_prompt = g.shape
        seq = np.repeat(np.arange(nseq, dtype=np.int32), n_prompt - 1)
        pos = np.tile(np.arange(n_prompt - 1, dtype=np.int32), nseq)
        tok = g[:, :-1].reshape(-1)
        for i in range(0, len(seq), rows_per_pass):
            out.append(("pass", gi, seq[i:i + rows_per_pass], pos[i:i + rows_per_pass], tok[i:i + rows_per_pass], False))
        out.append(("pass", gi, np.arange(nseq, dtype=np.int32), np.full(nseq, n_prompt - 1, np.int32), g[:, -1].copy(), True))
    for gi, g in enumerate(prompts):
        out.append(("decode", gi, int(np.asarray(g).shape[0]), n_steps))
    return out


class LibPipeline:
    """One LLM stage of a layer-sharded job, driven through the library's own hand-off (tk_mi355x_pipe_*: device mailboxes mapped with
    hipIpc, csrc/llm/tk_llm_pipe.h).  The host only (1) exchanges the 80-byte mailbox handles once, over `dist` (any backend: they are
    plain bytes), and (2) enqueues the same schedule on every stage; no tensor ever goes through torch.distributed and nothing
    synchronises with the host between a prompt and its last decoded token.

    make_pipe(group_index, stage, n_stages, l0, l1) -> an object with .handle_bytes, .connect(next_bytes, prev_bytes), .enqueue(seq, pos,
    tok, head), .decode(nrows, n_steps), .sync(nrows, n_steps); the GPU one wraps tk.LlmPipe (below), tests inject a recorder.
    rccl=True: the collective transport instead (SURVEY.md 8e's ncclSend / ncclRecv; bench.py --pipe-rccl): stage 0 makes one RCCL unique id
    per row group (.new_unique_id()), the 128-byte ids go round in the same exchange, every stage joins (.connect_rccl(id)); one GPU per stage."""

    def __init__(self, dist, llm_ranks, n_layer, n_groups, make_pipe, rccl=False):
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.llm_ranks = list(llm_ranks)
        self.n_stages = len(self.llm_ranks)
        self.stage = self.llm_ranks.index(self.rank) if self.rank in self.llm_ranks else None
        self.bounds = stage_bounds(n_layer, self.n_stages)
        self.pipes = []
        mine = []
        if self.stage is not None:
            for g in range(n_groups):
                self.pipes.append(make_pipe(g, self.stage, self.n_stages, self.bounds[self.stage], self.bounds[self.stage + 1]))
            mine = [p.handle_bytes for p in self.pipes]
            if rccl:  # the collective transport: stage 0 names one communicator per row group, everybody joins it
                mine = [p.new_unique_id() for p in self.pipes] if self.stage == 0 else []
        # every rank of the job takes part in the exchange (perception ranks contribute nothing), so any process group works
        if dist is not None and dist.get_world_size() > 1:
            table = [None] * dist.get_world_size()
            dist.all_gather_object(table, mine)
        else:
            table = [mine]
        if self.stage is not None and self.n_stages > 1 and rccl:
            ids = table[self.llm_ranks[0]]
            for g, p in enumerate(self.pipes):
                p.connect_rccl(ids[g])
        elif self.stage is not None and self.n_stages > 1:
            nxt = table[self.llm_ranks[(self.stage + 1) % self.n_stages]]
            prv = table[self.llm_ranks[(self.stage - 1) % self.n_stages]]
            for g, p in enumerate(self.pipes):
                p.connect(nxt[g], prv[g])

    @property
    def first(self):
        return self.stage == 0

    @property
    def last(self):
        return self.stage is not None and self.stage == self.n_stages - 1

    def generate(self, prompts, n_steps, rows_per_pass=256):
        """enqueue the whole job of this stage, then wait once.  Returns the per-group [n_steps][nseq] ids: sampled ones on the last
        stage, fed ones on stage 0, None on the others (and on ranks outside the pipeline)."""
        if self.stage is None:
            return None
        for item in stage_schedule(prompts, n_steps, rows_per_pass):
            if item[0] == "pass":
                _, g, seq, pos, tok, head = item
                self.pipes[g].enqueue(seq, pos, tok if self.first else None, head)
            else:
                _, g, nseq, steps = item
                self.pipes[g].decode(nseq, steps)
        out = []
        for g, p in enumerate(self.pipes):
            nseq = len(prompts[g])
            toks = p.sync(nseq, n_steps if (self.first or self.last) else 0)
            out.append(toks)
        return out if (self.first or self.last) else None


class _GpuPipe:
    """adapter of tk.LlmPipe to what LibPipeline drives"""

    def __init__(self, tk, session, stage, n_stages, l0, l1, payload_f16):
        self.tk = tk
        self.pipe = tk.LlmPipe(session, stage, n_stages, l0, l1, payload_f16=payload_f16)
        self.handle_bytes = self.pipe.handle.to_bytes()

    def connect(self, nxt, prv):
        self.pipe.connect(self.tk.PipeHandle.from_bytes(nxt), self.tk.PipeHandle.from_bytes(prv))

    def new_unique_id(self):
        return self.tk.LlmPipe.rccl_unique_id()

    def connect_rccl(self, unique_id):
        self.pipe.connect_rccl(unique_id)

    def enqueue(self, seq, pos, tok, head):
        self.pipe.enqueue(seq, pos, tok, head)

    def decode(self, nrows, n_steps):
        self.pipe.decode(nrows, n_steps)

    def sync(self, nrows, n_steps):
        return self.pipe.sync(nrows, n_steps)

    def close(self):
        self.pipe.close()


def gpu_pipe_factory(tk, model, rows_per_group, max_ctx, payload_f16=False):
    """make_pipe for LibPipeline on a GPU rank: one session (own stream, own KV cache) and one pipe per row group"""
    sessions = []

    def make(g, stage, n_stages, l0, l1):
        s = tk.LlmSession(model, rows_per_group, max_ctx)
        sessions.append(s)
        return _GpuPipe(tk, s, stage, n_stages, l0, l1, payload_f16)

    make.sessions = sessions
    return make
