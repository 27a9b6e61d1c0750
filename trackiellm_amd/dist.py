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


def perception_plan(vision_ranks, audio_ranks, cycles, tok_steps):
    """{rank: (frames, utterances, token ids per utterance)} a perception rank hands over per step: `cycles` split evenly (ceil) over the
    ranks of each stream — the shares bench.py gives its PerceptionBench objects.  Every rank computes the same table, so message sizes
    are known on both ends without a size exchange."""
    plan = {}
    for r in vision_ranks:
        f, u, t = plan.get(r, (0, 0, 0))
        plan[r] = (-(-cycles // len(vision_ranks)), u, t)
    for r in audio_ranks:
        f, u, t = plan.get(r, (0, 0, 0))
        plan[r] = (f, -(-cycles // len(audio_ranks)), tok_steps)
    return plan


def _message_words(frames, utts, tok_steps):
    return frames * (1 + MAX_DETS * 6) + utts * tok_steps


class PerceptionExchange:
    """The cycle's data dependency across ranks (model-per-gpu and combined placements): the perception ranks' results of a batch must be on
    the LLM's first rank (`dst`) before it generates for that batch — the software pipeline of the one-GPU bench (LLM of batch k beside the
    perception of batch k + 1) with its hand-off made explicit.

    Point to point and never blocking a sender on the LLM: messages have a FIXED size known to both ends from `plan`
    (perception_plan: per frame a count + MAX_DETS x 6 floats, per utterance tok_steps ids, as 32-bit words), a perception rank's
    hand_over() is one isend from one of two buffers (it only ever waits for its own send of two steps ago), dst's hand_over() posts the
    irecvs of the batch that has just been computed and returns, and require() — called right before generating — is the only wait.  A
    slow LLM rank therefore never serialises the perception ranks, and the perception ranks never wait inside a collective for ranks that
    have nothing to contribute (what one gather_object per step did).  `device`: where the message tensors live ("cuda" under the nccl
    backend = RCCL, None = host tensors for gloo)."""

    def __init__(self, dist, dst=0, plan=None, device=None):
        import torch
        self.dist, self.dst = dist, dst
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.plan = dict(plan or {})
        if self.dst in self.plan and any(self.plan[self.dst]):
            raise ValueError("the LLM's first rank cannot also be a perception rank of the exchange")
        self.received = None      # on dst: what the perception ranks produced for the batch last required
        self.bytes_last = 0
        self.checksum = 0
        self._torch = torch
        self._device = device
        self._step = 0
        self._sends = [None, None]
        self._posted = None       # on dst: [(rank, tensor, work)] of the newest batch
        mk = lambda n: torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        if self.rank == self.dst:
            self._rbuf = [{r: mk(_message_words(*self.plan[r])) for r in sorted(self.plan)} for _ in range(2)]
        elif self.rank in self.plan:
            self._sbuf = [mk(_message_words(*self.plan[self.rank])) for _ in range(2)]

    def _encode(self, payload):
        import numpy as np
        f, u, t = self.plan[self.rank]
        words = np.zeros(max(_message_words(f, u, t), 1), np.int32)
        payload = payload or {}
        if f:
            n = np.asarray(payload.get("n_dets", np.zeros(0, np.int32)), np.int32)
            d = np.asarray(payload.get("dets", np.zeros((0, MAX_DETS, 6), np.float32)), np.float32)
            if len(n) != f or d.shape != (f, MAX_DETS, 6):
                raise ValueError("this rank hands over %d frames per step (plan), got %d" % (f, len(n)))
            words[:f] = n
            words[f:f + f * MAX_DETS * 6] = d.reshape(-1).view(np.int32)
        if u:
            tk = np.asarray(payload.get("tokens", np.zeros((0, t), np.int32)), np.int32)
            if tk.shape != (u, t):
                raise ValueError("this rank hands over %d x %d token ids per step (plan), got %s" % (u, t, tk.shape))
            words[f * (1 + MAX_DETS * 6):f * (1 + MAX_DETS * 6) + u * t] = tk.reshape(-1)
        return words

    def _decode(self, rank, words):
        import numpy as np
        f, u, t = self.plan[rank]
        out = {}
        if f:
            out["n_dets"] = words[:f].copy()
            out["dets"] = words[f:f + f * MAX_DETS * 6].view(np.float32).reshape(f, MAX_DETS, 6).copy()
        if u:
            o = f * (1 + MAX_DETS * 6)
            out["tokens"] = words[o:o + u * t].reshape(u, t).copy()
        return out

    def hand_over(self, payload):
        """once per step on every rank, at the step's end.  Perception ranks: payload = pack_perception(...) of the batch just computed,
        sent without waiting for the receiver.  dst: posts the receives of that batch (payload ignored) and returns at once."""
        slot = self._step & 1
        self._step += 1
        if self.rank == self.dst:
            self._posted = [(r, self._rbuf[slot][r], self.dist.irecv(self._rbuf[slot][r], src=r)) for r in sorted(self.plan)]
            return None
        if self.rank not in self.plan:
            return None
        if self._sends[slot] is not None:
            self._sends[slot].wait()                       # the send of two steps ago: its buffer is reused now
            self._sends[slot] = None
        self._sbuf[slot].copy_(self._torch.from_numpy(self._encode(payload)))
        self._sends[slot] = self.dist.isend(self._sbuf[slot], dst=self.dst)
        return None

    def require(self, n_frames, n_utts):
        """on dst, before generating for a batch: wait for that batch's messages (posted by the previous hand_over) and check they are
        complete; the results become `received`"""
        import numpy as np
        if self.rank != self.dst:
            return None
        if self._posted is None:
            raise RuntimeError("perception results of this batch are missing: nothing has been handed over yet")
        got = []
        for r, buf, work in self._posted:
            work.wait()
            got.append(self._decode(r, buf.cpu().numpy()))
        self._posted = None
        self.received = got
        self.bytes_last = int(sum(v.nbytes for b in got for v in b.values()))
        cs = 0
        for b in got:  # consumed: every value is read once (the prompt builder's stand-in; the token ids of the prompt stay the fixed
            for k in sorted(b):  # seeded ones of SURVEY.md 8d, so the numbers compare with the one-GPU run)
                cs = (cs * 1000003 + int(np.asarray(b[k], np.float64).sum() * 16)) % (1 << 61)
        self.checksum = cs
        f = sum(len(b["n_dets"]) for b in got if "n_dets" in b)
        u = sum(len(b["tokens"]) for b in got if "tokens" in b)
        if f < n_frames or u < n_utts:
            raise RuntimeError("perception results of this batch are missing: %d / %d frames, %d / %d utterances" % (f, n_frames, u, n_utts))
        return got

    def finish(self):
        """before the process group goes away: a sender's last isends complete, dst takes the batch nobody generated for"""
        for w in self._sends:
            if w is not None:
                w.wait()
        self._sends = [None, None]
        if self.rank == self.dst and self._posted is not None:
            for _, _, work in self._posted:
                work.wait()
            self._posted = None


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
        if rccl and n_groups > 1 and self.n_stages > 1:
            # several communicators on one device with eager send / recv on separate streams is the pattern RCCL documents as
            # deadlock-prone; the collective transport carries ONE row group per stage (the mailbox transport has no such limit)
            raise ValueError("the RCCL transport carries one row group per pipeline (got %d): use --sessions 1 with --pipe-rccl" % n_groups)
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
