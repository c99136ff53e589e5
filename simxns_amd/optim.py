"""Fused clip_grad_norm_ + transformers.AdamW + zero_grad over the engines' flat f32 buffers, and
the linear warm-up schedule (SimANS/co_training/co_training_marco_train.py:57-69, 126-134, 246-254).

One ``step()`` = (optional RCCL all-reduce of the flat gradients) -> global L2 norm (device scalar, no
host sync) -> AdamW update of every tower with the clip coefficient computed on device -> gradients
zeroed in the same pass.  Non-engine parameters (the Reranker's Linear(H,1)) are packed into a small
extra flat buffer so that one code path serves all.
"""
import torch

from . import _lib as L


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
            self.state["extra"] = st
        return st

    def zero_grad(self, set_to_none=False):
        for m, e in self.towers:
            if e.flat_grad is not None:
                e.flat_grad.zero_()
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
        grad_scale = 1.0
        bufs = []
        for m, e in self.towers:
            bufs.append((e.flat, e.ensure_grad(), self._tower_state(e), e))
        if self.extra:
            st = self._extra_state(dev)
            o = 0
            for p in self.extra:
                n = p.numel()
                st["p"][o:o + n].copy_(p.reshape(-1))
                if p.grad is not None:
                    st["g"][o:o + n].copy_(p.grad.reshape(-1))
                o += n
            bufs.append((st["p"], st["g"], st, None))
        if world_size > 1:
            import torch.distributed as dist
            for p_, g_, st_, e_ in bufs:
                dist.all_reduce(g_, op=dist.ReduceOp.SUM, group=self.group)
            grad_scale = 1.0 / world_size
        sq = torch.zeros(1, dtype=torch.float32, device=dev)
        if max_grad_norm and max_grad_norm > 0:
            for p_, g_, st_, e_ in bufs:
                L.call("simx_sqnorm_accum", s, L.ptr(g_), g_.numel(), L.ptr(sq))
        for p_, g_, st_, e_ in bufs:
            L.call("simx_adamw_step", s, L.ptr(p_), L.ptr(g_), L.ptr(st_["m"]), L.ptr(st_["v"]), p_.numel(), lr,
                   self.betas[0], self.betas[1], self.eps, self.wd if e_ is None else 0.0, self.step_count,
                   L.ptr(sq) if max_grad_norm and max_grad_norm > 0 else None, float(max_grad_norm or 0.0), grad_scale, 1)
            if e_ is not None:
                e_.mark_weights_dirty()
        if self.wd > 0.0:
            self._decay_towers(lr)
        if self.extra:
            st = self.state["extra"]
            o = 0
            for p in self.extra:
                n = p.numel()
                p.copy_(st["p"][o:o + n].view_as(p))
                if p.grad is not None:
                    p.grad.zero_()
                o += n
        self._sq = sq
        return sq

    def _decay_towers(self, lr):
        # decoupled decay p -= lr*wd*p on everything except 'bias' / 'LayerNorm.weight'
        # (co_training_marco_train.py:59-65); default weight_decay is 0.0 so this is normally skipped.
        for m, e in self.towers:
            for name, p in m.named_parameters():
                if not any(nd in name for nd in ("bias", "LayerNorm.weight")):
                    p.mul_(1.0 - lr * self.wd)
            e.mark_weights_dirty()

    def state_dict(self):
        out = {"step": self.step_count, "lr": self.param_groups[0]["lr"], "towers": []}
        for m, e in self.towers:
            st = self._tower_state(e)
            out["towers"].append({"m": st["m"].cpu(), "v": st["v"].cpu()})
        if "extra" in self.state:
            out["extra"] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.state["extra"].items()}
        return out

    def load_state_dict(self, sd):
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
        return {"t": self.t, "warm": self.warm, "total": self.total, "base": self.base}

    def load_state_dict(self, sd):
        self.t, self.warm, self.total, self.base = sd["t"], sd["warm"], sd["total"], sd["base"]
        self._apply()
