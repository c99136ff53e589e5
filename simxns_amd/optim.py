"""Fused clip_grad_norm_ + transformers.AdamW + zero_grad over the engines' flat f32 buffers, and
the linear warm-up schedule (SimANS/co_training/co_training_marco_train.py:57-69, 126-134, 246-254).

One ``step()`` = (optional RCCL all-reduce of the flat gradients) -> global L2 norm (device scalar, no
host sync) -> AdamW update of every tower with the clip coefficient computed on device -> gradients
zeroed in the same pass.  Non-engine parameters (the Reranker's Linear(H,1)) are packed into a small
extra flat buffer so that one code path serves all.
"""
import torch

from . import _lib as L
from .engine import LossScaler


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _engines_of(model):
    seen, out = set(), []
    for m in model.modules():
        e = getattr(m, "engine", None)
        if e is not None and id(e) not in seen:
            seen.add(id(e))
            out.append((m, e))
    return out


class FusedAdamW(object):
    """AdamW(lr, eps, weight_decay) with the reference's no-decay grouping ('bias', 'LayerNorm.weight')."""

    def __init__(self, model, lr=1e-5, eps=1e-8, weight_decay=0.0, betas=(0.9, 0.999), process_group=None):
        self.model = model
        self.lr, self.eps, self.wd, self.betas = float(lr), float(eps), float(weight_decay), betas
        self.base_lr = float(lr)
        self.step_count = 0
        self.group = process_group
        self.towers = _engines_of(model)
        own = set()
        for m, e in self.towers:
            own.update(id(p) for p in m.parameters())
        self.extra = [p for p in model.parameters() if id(p) not in own and p.requires_grad]
        self.state = {}
        self.param_groups = [{"lr": self.lr}]           # scheduler-facing view
        self._sq = None
        # fp16 towers: this optimiser's dynamic loss scale (amp.initialize(model, optimizer, opt_level='O1') creates one
        # scaler per optimizer too, co_training_marco_train.py:97-104); engines read it in backward, step() updates it
        self.scaler = None
        if any(e.dtype_code == L.SIMX_F16 for m, e in self.towers):
            import os
            self.scaler = LossScaler(init_scale=float(os.environ.get("SIMX_LOSS_SCALE_INIT", 2.0 ** 16)))
            for m, e in self.towers:
                if e.dtype_code == L.SIMX_F16:
                    e.scaler = self.scaler
        # data parallelism (enable_overlap): gradient slices are all-reduced on a side stream as soon as the backward
        # has finished them; step() only waits for what is still in flight
        self.world_size = 1
        self.armed = True                                # False while accumulating micro-steps (no reduction yet)
        self.payload = "fp32"                            # "bf16": gradient slices cross the links as bf16 (half the bytes)
        self.profile_comm = False                        # bench.py --gpus N: per-step communication timings (comm_stats())
        self._comm = None
        self._hooks_on = False
        self._pending = []                               # [(work, engine, lo, hi, payload buffer or None, events or None)]
        self._reduced = {}                               # id(engine) -> backward serial whose slices were launched
        self._synced = False
        self._comm_log = []

    # ---- DDP's role: gradient averaging over the ranks (co_training_marco_train.py:107-114) ------------------------
    def enable_overlap(self, world_size, parts=2, process_group=None, payload=None, force=False):
        """One process per GPU, RCCL: every tower's backward runs in `parts` layer ranges (simx_bert_bwd_range) and each
        finished slice of the flat gradient buffer is all-reduced asynchronously on a communication stream while the
        remaining layers -- and the other tower -- are still in backward.  The query tower's reduction hides under the
        passage tower's backward, the upper half of the passage tower's under its lower half.  With gradient accumulation
        set ``armed = False`` for all but the last micro-step: an in-place accumulated buffer must be reduced ONCE, and a
        backward into a buffer whose slices are already on the wire raises (see _on_grad_ready).  `payload="bf16"` sends
        the slices as bf16 (half the bytes; the sum is taken in bf16 by RCCL, the master copy stays f32).
        `force`: install the hooks for a one-rank group as well (the RCCL smoke test of the whole path on one GPU)."""
        self.world_size = int(world_size)
        if process_group is not None:
            self.group = process_group
        if payload is None:
            # default: the payload follows the engine's arithmetic.  f32 engines (every shipped recipe; the mode held to
            # north_star's 1e-3) all-reduce f32 like the reference's DDP (co_training_marco_train.py:107-114); only when EVERY
            # tower computes in 16 bits do the slices cross the links as bf16 (438 MB instead of 876 MB per step and
            # rank; the sum is taken in bf16 by RCCL, the master gradients stay f32).  SIMX_GRAD_PAYLOAD=bf16|fp32 overrides.
            import os
            all16 = bool(self.towers) and all(getattr(e, "dtype_code", None) in (L.SIMX_BF16, L.SIMX_F16) for _, e in self.towers)
            payload = os.environ.get("SIMX_GRAD_PAYLOAD") or ("bf16" if self.world_size > 1 and all16 else "fp32")
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16', got %r" % (payload,))
        self.payload = payload
        if self.world_size <= 1 and not force:
            return self
        dev = self.towers[0][1].flat.device if self.towers else None
        self._comm = torch.cuda.Stream(device=dev) if dev is not None and dev.type == "cuda" else None
        self._hooks_on = True
        for m, e in self.towers:
            e.grad_ready_hook = self._on_grad_ready
            e.bwd_parts = parts
        return self

    def _on_grad_ready(self, e, lo, hi):
        if not self.armed or not self._hooks_on or hi <= lo:
            return
        import torch.distributed as dist
        self._reduced[id(e)] = e._bwd_serial
        e._reduced_this_step = True                      # a further backward into this buffer before step() raises
        sl = e.flat_grad[lo:hi]
        if e.flat_grad.is_cuda:
            cur = torch.cuda.current_stream()
            comm = self._comm or cur
            comm.wait_stream(cur)                        # the slice is final once the backward kernels queued so far are done
            with torch.cuda.stream(comm):
                ev = None
                if self.profile_comm:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(comm)
                buf = sl.to(torch.bfloat16) if self.payload == "bf16" else None
                w = dist.all_reduce(buf if buf is not None else sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:                                            # (host tensors: the gloo control-flow tests)
            ev = None
            buf = sl.to(torch.bfloat16) if self.payload == "bf16" else None
            w = dist.all_reduce(buf if buf is not None else sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((w, e, lo, hi, buf, ev))

    def sync_grads(self, world_size=None):
        """Completes the gradient reduction of this step (idempotent until the next step()): joins the streams the towers'
        backward ran on, waits for the slices launched by the backward hooks, reduces whatever was not launched (no hooks,
        un-armed micro-steps, the small non-engine parameters) synchronously.  -> the factor that turns the summed
        gradients into the DDP mean (1/W)."""
        W = int(world_size if world_size is not None else self.world_size)
        if W <= 1 and not self._hooks_on:
            return 1.0
        if self._synced:
            return 1.0 / W
        import torch.distributed as dist
        on_gpu = bool(self.towers) and self.towers[0][1].flat.is_cuda
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu:
            for m, e in self.towers:                   # a tower's backward may have run on a side stream (BiBertEncoder):
                st = getattr(e, "last_stream", None)   # no collective may read its buffer before that stream is done
                if st is not None and st != cur:
                    cur.wait_stream(st)
        launched = set()
        comm = self._comm if on_gpu else None
        nbytes, evs = 0, []
        ctx = torch.cuda.stream(comm) if comm is not None else _NullCtx()
        with ctx:
            for w, e, lo, hi, buf, ev in self._pending:
                w.wait()                                  # RCCL: the communication stream waits for the collective, not the host
                if buf is not None:
                    e.flat_grad[lo:hi].copy_(buf)
                if ev is not None:
                    ev[1].record(comm)
                    evs.append(ev)
                launched.add(id(e))
                nbytes += (hi - lo) * (2 if buf is not None else 4)
        self._pending = []
        t_wait = None
        if comm is not None:
            if self.profile_comm:
                t_wait = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                t_wait[0].record(cur)
            cur.wait_stream(comm)
            if t_wait is not None:
                t_wait[1].record(cur)
        for m, e in self.towers:
            if id(e) not in launched:
                g = e.ensure_grad()
                if self.payload == "bf16":
                    b = g.to(torch.bfloat16)
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                    g.copy_(b)
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                nbytes += g.numel() * (2 if self.payload == "bf16" else 4)
            e._open_graphs = 0
        if self.extra:
            st = self._extra_state(self.extra[0].device)
            self._gather_extra(st)
            dist.all_reduce(st["g"], op=dist.ReduceOp.SUM, group=self.group)
            st["g_ready"] = True
            nbytes += st["g"].numel() * 4
        if self.profile_comm:
            self._comm_log.append((nbytes, evs, t_wait, len(launched)))
        self._synced = True
        return 1.0 / W

    def comm_stats(self):
        """Per-step communication figures collected while ``profile_comm`` was on (synchronises): bytes this rank put on the
        wire, milliseconds the all-reduces occupied the communication stream, milliseconds the compute stream had to wait
        for them in step() (the exposed part), slices launched from the backward hooks."""
        if not self._comm_log:
            return None
        torch.cuda.synchronize()
        n = len(self._comm_log)
        tot_b = sum(b for b, _, _, _ in self._comm_log)
        ar = sum(sum(a.elapsed_time(b) for a, b in evs) for _, evs, _, _ in self._comm_log)
        ex = sum(t[0].elapsed_time(t[1]) for _, _, t, _ in self._comm_log if t is not None)
        return {"steps": n, "allreduce_bytes_per_step": tot_b // n, "payload": self.payload,
                "allreduce_ms_on_comm_stream_per_step": round(ar / n, 3), "exposed_wait_ms_per_step": round(ex / n, 3),
                "overlapped_slices_per_step": sum(k for _, evs, _, k in self._comm_log) / float(n)}

    def _gather_extra(self, st):
        o = 0
        for p in self.extra:
            n = p.numel()
            st["p"][o:o + n].copy_(p.reshape(-1))
            if p.grad is not None:
                st["g"][o:o + n].copy_(p.grad.reshape(-1))
            o += n

    def _tower_state(self, e):
        st = self.state.get(id(e))
        if st is None or st["m"].device != e.flat.device:
            st = {"m": torch.zeros_like(e.flat), "v": torch.zeros_like(e.flat)}
            self.state[id(e)] = st
        return st

    def _extra_state(self, dev):
        st = self.state.get("extra")
        n = sum(p.numel() for p in self.extra)
        if st is None or st["p"].device != dev:
            pad = (n + 3) // 4 * 4
            st = {k: torch.zeros(pad, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
            st["n"] = n
            st["g_ready"] = False
            self.state["extra"] = st
        return st

    def _discard_pending(self):
        """Drop a step whose gradient reduction is already on the wire (zero_grad() without step(), e.g. after a non-finite
        loss): the collectives launched by the backward hooks are waited for on the communication stream, the compute stream
        joins it -- nothing may zero a buffer RCCL is still reading or writing -- and the per-step bookkeeping is reset so
        that the next backward may arm its slices again."""
        if self._pending:
            on_gpu = bool(self.towers) and self.towers[0][1].flat.is_cuda
            comm = self._comm if on_gpu else None
            ctx = torch.cuda.stream(comm) if comm is not None else _NullCtx()
            with ctx:
                for w, e, lo, hi, buf, ev in self._pending:
                    w.wait()
            if comm is not None:
                torch.cuda.current_stream().wait_stream(comm)
            self._pending = []
        self._reduced = {}
        self._synced = False
        for m, e in self.towers:
            e._reduced_this_step = False
        st = self.state.get("extra")
        if st is not None:
            st["g_ready"] = False

    def zero_grad(self, set_to_none=False):
        self._discard_pending()
        for m, e in self.towers:
            if e.flat_grad is not None:
                e.flat_grad.zero_()
            e._open_graphs = 0                         # forwards that never got a backward no longer hold the hooks back
        for p in self.extra:
            if p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, max_grad_norm=0.0, world_size=1):
        """Returns the device scalar holding the squared global grad norm (before clipping)."""
        lr = self.param_groups[0]["lr"]
        self.step_count += 1
        dev = self.towers[0][1].flat.device if self.towers else self.extra[0].device
        cur = torch.cuda.current_stream()
        for m, e in self.towers:                       # a tower's backward may have run on a side stream (BiBertEncoder)
            st = getattr(e, "last_stream", None)
            if st is not None and st != cur:
                cur.wait_stream(st)
        s = L.stream_ptr()
        grad_scale = self.sync_grads(max(int(world_size), self.world_size))
        self._synced = False
        self._reduced = {}
        bufs = []
        for m, e in self.towers:
            bufs.append((e.flat, e.ensure_grad(), self._tower_state(e), e))
            e._open_graphs = 0
            e._reduced_this_step = False
        if self.extra:
            st = self._extra_state(dev)
            if not st.get("g_ready"):
                self._gather_extra(st)
            st["g_ready"] = False
            bufs.append((st["p"], st["g"], st, None))
        sq = torch.zeros(1, dtype=torch.float32, device=dev)
        clip = bool(max_grad_norm and max_grad_norm > 0)
        if clip or self.scaler is not None:       # (the fp16 overflow check is the same norm: inf / nan in any gradient)
            ws = torch.empty(2048, dtype=torch.float32, device=dev)
            for p_, g_, st_, e_ in bufs:          # (no float atomics: every data-parallel replica gets the same bits)
                L.call("simx_sqnorm_accum_det", s, L.ptr(g_), g_.numel(), L.ptr(sq), L.ptr(ws))
        sc = None
        if self.scaler is not None:
            self.scaler.update(sq)                # device side: skip + halve on overflow, count, grow
            sc = self.scaler.state(dev)
        for p_, g_, st_, e_ in bufs:
            L.call("simx_adamw_step_sc", s, L.ptr(p_), L.ptr(g_), L.ptr(st_["m"]), L.ptr(st_["v"]), p_.numel(), lr,
                   self.betas[0], self.betas[1], self.eps, 0.0, self.step_count,
                   L.ptr(sq) if clip else None, float(max_grad_norm or 0.0), grad_scale, 1, L.ptr(sc))
            if e_ is not None:
                e_.mark_weights_dirty()
        # (on an overflow-skipped fp16 step the decay factor below is 1: the device flag, no host read)
        keep = None if sc is None else 1.0 - sc[3]
        if self.wd > 0.0:
            self._decay_towers(lr, keep)
        if self.extra:
            st = self.state["extra"]
            nodecay = self._extra_no_decay()
            o = 0
            for p in self.extra:
                n = p.numel()
                p.copy_(st["p"][o:o + n].view_as(p))
                if self.wd > 0.0 and id(p) not in nodecay:      # same grouping rule as the towers (:59-65)
                    p.mul_(1.0 - lr * self.wd if keep is None else 1.0 - lr * self.wd * keep)
                if p.grad is not None:
                    p.grad.zero_()
                o += n
        self._sq = sq
        return sq

    def _extra_no_decay(self):
        return {id(p) for n, p in self.model.named_parameters() if any(nd in n for nd in ("bias", "LayerNorm.weight"))}

    def _decay_towers(self, lr, keep=None):
        # decoupled decay p -= lr*wd*p on everything except 'bias' / 'LayerNorm.weight'
        # (co_training_marco_train.py:59-65); default weight_decay is 0.0 so this is normally skipped.
        f = 1.0 - lr * self.wd if keep is None else 1.0 - lr * self.wd * keep
        for m, e in self.towers:
            for name, p in m.named_parameters():
                if not any(nd in name for nd in ("bias", "LayerNorm.weight")):
                    p.mul_(f)
            e.mark_weights_dirty()

    # ---- state interchange with the reference's checkpoints (co_training_marco_train.py:310-358) ------------------------
    def _ref_param_order(self):
        """The reference builds transformers.AdamW from two groups in named_parameters() order -- decayed parameters first,
        then the 'bias' / 'LayerNorm.weight' ones (co_training_marco_train.py:57-65) -- and torch numbers optimizer state by
        position in that concatenation.  -> [(name, param, group)]"""
        named = list(self.model.named_parameters())
        nd = lambda n: any(k in n for k in ("bias", "LayerNorm.weight"))
        return [(n, p, 0) for n, p in named if not nd(n)] + [(n, p, 1) for n, p in named if nd(n)]

    def _moment_views(self):
        """{id(param): (m_view, v_view)} into the flat Adam-state buffers."""
        out = {}
        for m, e in self.towers:
            st = self._tower_state(e)
            mv, vv = e.views(st["m"]), e.views(st["v"])
            for name, p in m.named_parameters():
                out[id(p)] = (mv[name], vv[name])
        if self.extra:
            st = self._extra_state(self.extra[0].device)
            o = 0
            for p in self.extra:
                n = p.numel()
                out[id(p)] = (st["m"][o:o + n].view_as(p), st["v"][o:o + n].view_as(p))
                o += n
        return out

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() of the reference's transformers.AdamW: {'state': {i: {'step', 'exp_avg',
        'exp_avg_sq'}}, 'param_groups': [decay group, no-decay group]} -- a checkpoint written here resumes in the reference
        and the other way round."""
        order = self._ref_param_order()
        views = self._moment_views()
        state = {}
        # (fp16: steps the scaler skipped did not age the moments -- apex skips optimizer.step() on them as well)
        applied = self.step_count if self.scaler is None else self.scaler.snapshot()["applied_steps"]
        if applied > 0:
            for i, (n, p, g) in enumerate(order):
                mv, vv = views[id(p)]
                state[i] = {"step": applied, "exp_avg": mv.detach().cpu().clone(), "exp_avg_sq": vv.detach().cpu().clone()}
        lr = self.param_groups[0]["lr"]
        base = dict(lr=lr, initial_lr=self.base_lr, betas=tuple(self.betas), eps=self.eps, correct_bias=True)
        n0 = sum(1 for _, _, g in order if g == 0)
        out = {"state": state,
               "param_groups": [dict(base, weight_decay=self.wd, params=list(range(n0))),
                                dict(base, weight_decay=0.0, params=list(range(n0, len(order))))]}
        if self.scaler is not None:
            out["loss_scaler"] = self.scaler.state_dict()     # (extra key: torch's Optimizer.load_state_dict ignores it)
        return out

    def load_state_dict(self, sd):
        if "towers" in sd:                              # round-1 format of this package (flat buffers)
            self.step_count = sd["step"]
            self.param_groups[0]["lr"] = sd.get("lr", self.lr)
            for (m, e), t in zip(self.towers, sd["towers"]):
                st = self._tower_state(e)
                st["m"].copy_(t["m"])
                st["v"].copy_(t["v"])
            if "extra" in sd and self.extra:
                st = self._extra_state(self.extra[0].device)
                for k in ("p", "g", "m", "v"):
                    st[k].copy_(sd["extra"][k])
            return
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("optimizer state is neither a torch Optimizer state_dict nor this package's flat format: keys %s"
                             % sorted(sd.keys()))
        order = self._ref_param_order()
        n_saved = sum(len(g["params"]) for g in sd["param_groups"])
        if n_saved != len(order):
            raise ValueError("optimizer state has %d parameters, this model has %d (different architecture or share_weight)"
                             % (n_saved, len(order)))
        views = self._moment_views()
        steps = set()
        with torch.no_grad():
            for i, (n, p, g) in enumerate(order):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %d (%s): shape %s vs parameter %s" % (i, n, tuple(st["exp_avg"].shape), tuple(p.shape)))
                mv, vv = views[id(p)]
                mv.copy_(st["exp_avg"].to(torch.float32))
                vv.copy_(st["exp_avg_sq"].to(torch.float32))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): the fused update keeps one step count" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        if self.scaler is not None:
            dev = self.towers[0][1].flat.device if self.towers else self.extra[0].device
            ls = dict(sd.get("loss_scaler") or {"scale": self.scaler.init_scale})
            ls["applied_steps"] = self.step_count
            self.scaler.load_state_dict(ls, dev)
        g0 = sd["param_groups"][0]
        self.param_groups[0]["lr"] = float(g0.get("lr", self.lr))
        self.base_lr = float(g0.get("initial_lr", self.base_lr))


class LinearWarmupSchedule(object):
    """get_linear_schedule_with_warmup (co_training_marco_train.py:126-134): lr*t/warm, then linear to 0."""

    def __init__(self, optimizer, num_warmup_steps, num_training_steps, last_step=0):
        self.opt, self.warm, self.total = optimizer, float(num_warmup_steps), float(num_training_steps)
        self.base = optimizer.base_lr
        self.t = last_step
        self._apply()

    def factor(self, t):
        if t < self.warm:
            return float(t) / float(max(1.0, self.warm))
        return max(0.0, float(self.total - t) / float(max(1.0, self.total - self.warm)))

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.base * self.factor(self.t)

    def step(self):
        self.t += 1
        self._apply()

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]

    def state_dict(self):
        """torch LambdaLR.state_dict() (what the reference checkpoints as scheduler_dict; lr_lambdas holds None for a plain
        function) -- last_epoch is the number of scheduler.step() calls."""
        return {"base_lrs": [self.base], "last_epoch": int(self.t), "_step_count": int(self.t) + 1, "verbose": False,
                "_get_lr_called_within_step": False, "_last_lr": self.get_last_lr(), "lr_lambdas": [None],
                "num_warmup_steps": self.warm, "num_training_steps": self.total}

    def load_state_dict(self, sd):
        if "last_epoch" in sd:                          # torch LambdaLR (the reference) or this class
            self.t = int(sd["last_epoch"])
            if sd.get("base_lrs"):
                self.base = float(sd["base_lrs"][0])
            self.warm = float(sd.get("num_warmup_steps", self.warm))       # (a LambdaLR dict does not carry the schedule shape:
            self.total = float(sd.get("num_training_steps", self.total))   #  it stays what the constructor was given)
        elif "t" in sd:                                 # round-1 format: written when the train loop still called
            # scheduler.step() BEFORE optimizer.step(), i.e. `t` had already been advanced for the update it was saved after;
            # under today's order (optimizer first, as the reference) the same next learning rate needs t + 1
            import warnings
            warnings.warn("LinearWarmupSchedule: loading a round-1-format scheduler state ('t'): it is read as written by the "
                          "scheduler-before-optimizer train loop of that round (t + 1); checkpoints written since carry 'last_epoch'")
            self.t, self.warm, self.total, self.base = sd["t"] + 1, sd["warm"], sd["total"], sd["base"]
        else:
            raise ValueError("scheduler state has neither 'last_epoch' nor 't': keys %s" % sorted(sd.keys()))
        self._apply()
