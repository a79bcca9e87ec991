"""Fused SM3 training step (the path bench.py times and tools/backbone_train.py drives by default):

    zero grads -> forward (4 encoder passes, 4+ projector passes) -> fused NT-Xent loss + d(z) for every
    loss term -> backward -> (data parallel: bucketed gradient all-reduce over RCCL, overlapped with the
    remaining backward) -> fused AdamW over the flat parameter buffer

Equivalent to tools/backbone_train.py:98-127 of the reference (loss composition :99-121, AdamW :525-527);
no autograd graph, no logits tensors, no host synchronisation inside the step.
"""
import os

import torch
import torch.distributed as dist

from . import ops
from .bridge import sm3_engine_for


class SM3Trainer:
    def __init__(self, model, lr, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999), style=0, data_parallel=None,
                 sync_bn=None, loss_scale=None, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=2000, global_negatives=False, target_momentum=None):
        """loss_scale: None = on exactly when the model's arithmetic is fp16 (the reference's AMP recipe: autocast +
        GradScaler with its defaults, tools/backbone_train.py:98,125-127,480); True / False force it.
        global_negatives (needs world * 2B and proj_dim to be multiples of 32): every NT-Xent term contrasts this rank's 2B
        projections against the projections of ALL ranks,
        all-gathered over RCCL (BASELINE.json north_star).  NOT the reference's behaviour -- its negatives are the local
        batch and more negatives mean a larger logsumexp -- hence opt-in, default off.
        target_momentum: None (reference behaviour: both views through the online network) or m in [0, 1): north_star's
        "momentum-updated target encoders", an extension with no counterpart in the reference.  A target copy of ALL
        parameters follows the online ones by EMA after every step (sm3_ema_update); each loss term then pairs the
        online projection of one view with the target projection (no gradient) of the other, symmetrised:
        L = 1/2 [NTXent(cat(q_v0, k_v1)) + NTXent(cat(k_v0, q_v1))], q online, k target (MoCo-v3 / BYOL style)."""
        self.model = model
        self.kind = model._KIND
        self.lr, self.wd, self.eps, self.betas, self.style = lr, weight_decay, eps, betas, style
        self.step_count = 0
        self.m = self.v = None
        self.dp = data_parallel if data_parallel is not None else (dist.is_available() and dist.is_initialized()
                                                                   and dist.get_world_size() > 1)
        self.sync_bn = self.dp if sync_bn is None else sync_bn
        self.world = dist.get_world_size() if self.dp else 1
        self._handles = []
        self._pending = []
        self.loss = None
        self.global_negatives = bool(global_negatives)
        self.target_momentum = target_momentum
        self.flat_target = None
        self.loss_scale = loss_scale
        self.scaler_cfg = (float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval))
        self._scaler = None  # device state: scale, found_inf, growth tracker, optimizer steps taken

    # ---- loss weights: tools/backbone_train.py:99-121 -----------------------------------
    def _weights(self, names):
        w = {}
        ncross = sum(1 for n in names if n.startswith("cross"))
        for n in names:
            w[n] = (0.25 if self.style == 2 else 0.5) if n.startswith("cross") else 1.0
        assert ncross in (0, 2, 4)
        return w

    def _engine(self):
        eng = sm3_engine_for(self.model, self.kind)
        if self.dp and getattr(self, "_groups", None) is None:
            # One communicator per execution lane (derm / clinic run on two streams): collectives of one
            # communicator execute in enqueue order, so with a single one the clinic lane's first statistics
            # all-reduce would queue behind ALL of the derm lane's and the lanes would serialise.  Every rank
            # creates the groups in the same order; each lane's call sequence is identical on every rank.
            # Gradient buckets get a communicator of their own: on a lane's communicator a 60 MB bucket would sit
            # in front of that lane's next (latency-critical) BatchNorm statistics all-reduce.
            lanes = list(eng.branches) + [k + "#1" for k in eng.branches] + ["main", "grads"]
            self._groups = {k: dist.new_group() for k in lanes}
        if self.dp and self.sync_bn and eng.__dict__.get("_explicit_sync") is None:
            eng.world_size = self.world
            if os.environ.get("SM3_SYNCBN_P2P", "0") == "1":
                # opt-in: the statistics exchange as one kernel on the lane's own stream through hipIpc-mapped mailboxes
                # (csrc/p2p.hip) instead of an RCCL all-reduce; single node, at most 8 ranks
                from .p2p import P2PStatSync
                dev = eng.store.flat_p.device if eng.store is not None else torch.device("cuda", torch.cuda.current_device())
                self._p2p = P2PStatSync([k for k in self._groups if k != "grads"], dev,
                                        timeout_s=float(os.environ.get("SM3_P2P_TIMEOUT_S", "20")))
                eng.stat_sync = lambda t: self._p2p(eng._lane, t)
            else:
                eng.stat_sync = lambda t: dist.all_reduce(t, group=self._groups[eng._lane])
            eng.__dict__["_explicit_sync"] = True
        return eng

    def _bucket_ready(self, eng, first, last):
        """Gradients of the parameters whose names start with first..last are final: all-reduce that slice of
        the flat gradient buffer now, on RCCL's stream, while backward continues."""
        st = eng.store
        a, b = self._bucket_range(eng, first, last)
        self._handles.append(dist.all_reduce(st.flat_g[a:b], async_op=True, group=self._groups["grads"]))
        self._handle_ranges.append((a, b))

    def _bucket_range(self, eng, first, last):
        st = eng.store
        names = st.names
        lo = next(i for i, n in enumerate(names) if n.startswith(first))
        hi = max(i for i, n in enumerate(names) if n.startswith(last))
        a = st.offsets[names[lo]]
        b = st.offsets[names[hi]] + (st._view(st.flat_g, names[hi]).numel() + 15) // 16 * 16
        return a, min(b, st.total)

    def _bucket_adamw(self, eng, first, last):
        """AdamW over the slice of the flat buffers whose gradients just became final (current = that lane's stream)."""
        st = eng.store
        a, b = self._bucket_range(eng, first, last)
        ops.adamw(st.flat_p[a:b], st.flat_g[a:b], self.m[a:b], self.v[a:b], self.lr, self.betas[0], self.betas[1], self.eps,
                  self.wd, self.step_count, 1.0)

    # ---- global negatives: all-gather of the projection embeddings (north_star; opt-in) ---------------------------
    def _ntxent_global(self, eng, name, z, T, weight, loss, dz_out, dz_scale):
        """One NT-Xent term with this rank's rows as anchors and the rows of every rank as candidates.  Exchanges: an
        all-gather of the normalised projections [2B, D] and an all-reduce of the candidate-role gradient [world*2B, D],
        both fp32 over the "main" communicator.  The similarity and its two gradient products run on the exact-f32 MFMA
        gather-GEMM / weight-gradient kernels (a few MFLOP)."""
        from ._lib import SM3_F32
        dev = z.device
        Rl, D = z.shape
        W = self.world if self.dp else 1
        rank = dist.get_rank() if self.dp else 0
        Rg = W * Rl
        if Rg % 32 or D % 32:  # K chunk of the exact-f32 gather-GEMM: S = zn zg^T sums over D, dS zg over Rg
            raise ValueError(f"global_negatives needs world * 2B (= {Rg}) and proj_dim (= {D}) to be multiples of 32")
        zn = torch.empty_like(z)
        inv = torch.empty(Rl, dtype=torch.float32, device=dev)
        ops.normalize_rows(z, zn, inv)
        if W > 1:
            zg = torch.empty(Rg, D, dtype=torch.float32, device=dev)
            dist.all_gather(list(zg.view(W, Rl, D).unbind(0)), zn, group=self._groups["main"])
        else:
            zg = zn
        S = torch.empty(Rl, Rg, dtype=torch.float32, device=dev)
        ops.conv_gemm(ops.fwd_desc(SM3_F32, Rl, 1, 1, D, Rg, 1, 1, 0), zn, zg, S, None, None)       # S = zn zg^T
        ops.ntxent_rect(S, rank * Rl, T, weight, loss, dz_scale=dz_scale)                           # S <- dloss/dS
        dzn_anchor = torch.empty(Rl, D, dtype=torch.float32, device=dev)
        zgt = zg.t().contiguous()                                                                    # [D, Rg]
        ops.conv_gemm(ops.fwd_desc(SM3_F32, Rl, 1, 1, Rg, D, 1, 1, 0), S, zgt, dzn_anchor, None, None)  # dS zg
        dzg = torch.zeros(Rg, D, dtype=torch.float32, device=dev)
        ops.conv_wgrad(ops.fwd_desc(SM3_F32, Rl, 1, 1, D, Rg, 1, 1, 0), zn, S, dzg)                  # dS^T zn
        if W > 1:
            dist.all_reduce(dzg, group=self._groups["main"])
        ops.normalize_rows_bwd(eng.dtype, dzn_anchor, dzg[rank * Rl:(rank + 1) * Rl], zn, inv, dz_out)

    # ---- dynamic loss scaling (fp16): torch.cuda.amp.GradScaler with its state on the device ----------------
    def _scaler_state(self, dev, dtype):
        on = self.loss_scale if self.loss_scale is not None else (dtype == torch.float16)
        if not on:
            return None
        if self._scaler is None or self._scaler["scale"].device != dev:
            # a growth tracker loaded before the first step (resume: load_scaler_state_dict runs when no device state
            # exists yet) is carried over, as GradScaler.load_state_dict does
            tracker0 = self.__dict__.pop("_tracker_init", 0)
            self._scaler = {"scale": torch.full((1,), self.scaler_cfg[0], dtype=torch.float32, device=dev),
                            "found_inf": torch.zeros(1, dtype=torch.int32, device=dev),
                            "tracker": torch.full((1,), tracker0, dtype=torch.int32, device=dev),
                            "steps": torch.full((1,), self.step_count, dtype=torch.int32, device=dev)}
        return self._scaler

    def scaler_state_dict(self):
        """The "scaler" entry of the reference's checkpoint (GradScaler.state_dict(), backbone_train.py:586)."""
        if self._scaler is None:
            return {}
        return {"scale": float(self._scaler["scale"]), "growth_factor": self.scaler_cfg[1],
                "backoff_factor": self.scaler_cfg[2], "growth_interval": self.scaler_cfg[3],
                "_growth_tracker": int(self._scaler["tracker"])}

    def load_scaler_state_dict(self, sd):
        if not sd:
            return
        self.scaler_cfg = (float(sd["scale"]), float(sd["growth_factor"]), float(sd["backoff_factor"]),
                           int(sd["growth_interval"]))
        if self._scaler is not None:
            self._scaler["scale"].fill_(float(sd["scale"]))
            self._scaler["tracker"].fill_(int(sd.get("_growth_tracker", 0)))
        else:
            self._tracker_init = int(sd.get("_growth_tracker", 0))

    def step(self, derm_imgs, clinic_imgs, metadata=None):
        """One optimizer step on this rank's batch; returns the (device, fp32, 1-element) loss tensor.
        metadata (extension; model built with metadata_dim): [B, d] fp32 -- two more NT-Xent terms of weight 1/2 contrast
        meta_proj(metadata) with the cross-modal projections of the first derm / clinic views."""
        with ops.stream_scope():  # launches outside the lanes go to the stream that is current now
            loss = self._step(derm_imgs, clinic_imgs, metadata)
            p2p = self.__dict__.get("_p2p")
            if p2p is not None:
                # a statistics exchange that timed out: the kernel has poisoned its sums (this step's loss is NaN), and
                # the flag raises here no later than the next step -- without a host wait inside the step
                try:
                    p2p.poll()
                except RuntimeError:
                    self._sync_step_count()
                    raise
            return loss

    def check(self):
        """Raise if a peer-to-peer statistics exchange has failed so far (synchronises; no-op on the RCCL path)."""
        p2p = self.__dict__.get("_p2p")
        if p2p is not None:
            try:
                p2p.check()
            except RuntimeError:
                self._sync_step_count()
                raise

    def _sync_step_count(self):
        """After a failed peer-to-peer exchange: the host's step counter back to the optimizer steps really APPLIED (the device
        counts them, `_p2p_taken`), so that a checkpoint written by a caller who catches the error resumes with the bias
        correction of the last good step."""
        taken = self.__dict__.get("_p2p_taken")
        if taken is not None:
            self.step_count = int(taken.item())

    def close(self, barrier=True):
        """Release the peer-to-peer mailboxes (SM3_SYNCBN_P2P=1); safe to call more than once."""
        p2p = self.__dict__.pop("_p2p", None)
        if p2p is not None:
            eng = sm3_engine_for(self.model, self.kind)
            eng.stat_sync = None
            eng.__dict__["_explicit_sync"] = None
            p2p.close(barrier=barrier)

    def __del__(self):
        try:
            self.close(barrier=False)
        except Exception:  # interpreter teardown: the device, the library or the process group may be gone already
            pass

    def _step(self, derm_imgs, clinic_imgs, metadata=None):
        eng = self._engine()
        dev = derm_imgs[0].device
        eng.prepare(dev)
        st = eng.store
        if self.m is None or self.m.numel() != st.total or self.m.device != dev:
            self.m = torch.zeros(st.total, dtype=torch.float32, device=dev)
            self.v = torch.zeros(st.total, dtype=torch.float32, device=dev)
        st.flat_g.zero_()
        self.model.train()
        views = {"derm": list(derm_imgs), "clinic": list(clinic_imgs)}
        zt = None
        if self.target_momentum is not None:
            # keys: the momentum target's projections of the same batch (no gradient, BatchNorm buffers untouched)
            if self.flat_target is None or self.flat_target.numel() != st.total:
                self.flat_target = st.flat_p.clone()
            online, st.flat_p = st.flat_p, self.flat_target
            eng.__dict__["_no_stat_update"] = True
            try:
                zt, _f, _s = eng.forward(views, self.style, True, False)
            finally:
                st.flat_p = online
                eng.__dict__["_no_stat_update"] = False
        if metadata is not None and (self.style != 0 or self.target_momentum is not None):
            raise NotImplementedError("the metadata branch is defined for style 0 without a momentum target")
        zs, _feats, saved = eng.forward(views, self.style, True, True, metadata=metadata)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        z_meta = zs.pop("meta", None)
        weights = self._weights(list(zs))
        dz = {}
        T = float(self.model.temperature)
        sc = self._scaler_state(dev, eng.tdt)
        batched = False
        if zt is None and not self.global_negatives and os.environ.get("SM3_NTXENT_BATCH", "1") != "0" and dev.type == "cuda":
            # the step's terms (equal shapes) in three launches instead of three per term, same bits (sm3_ntxent_fused_batch)
            names = list(zs)
            shp = zs[names[0]].shape
            ws = eng._work("ntxent_ws_batch", ops.ntxent_batch_workspace_floats(len(names), *shp))
            for name in names:
                dz[name] = torch.empty(shp[0], shp[1], dtype=eng.tdt, device=dev)
            batched = ops.ntxent_fused_batch(eng.dtype, [zs[n] for n in names], T, [weights[n] for n in names], ws, loss,
                                             [dz[n] for n in names], dz_scale=sc["scale"] if sc is not None else None)
        for name, z in (() if batched else zs.items()):
            R, D = z.shape
            ws = eng._work("ntxent_ws", ops.ntxent_workspace_floats(R, D))
            dz[name] = torch.empty(R, D, dtype=eng.tdt, device=dev)
            scale = sc["scale"] if sc is not None else None
            if zt is not None:
                # symmetrised query/key terms; only the query half of each gradient is kept
                h = R // 2
                k = zt[name]
                for q_first in (True, False):
                    zz = torch.cat([z[:h], k[h:]] if q_first else [k[:h], z[h:]], 0)
                    dd = torch.empty(R, D, dtype=eng.tdt, device=dev)
                    if self.global_negatives:
                        self._ntxent_global(eng, name, zz, T, 0.5 * weights[name], loss, dd, scale)
                    else:
                        ops.ntxent_fused(eng.dtype, zz, T, 0.5 * weights[name], ws, loss, dd, dz_scale=scale)
                    if q_first:
                        dz[name][:h].copy_(dd[:h])
                    else:
                        dz[name][h:].copy_(dd[h:])
            elif self.global_negatives:
                self._ntxent_global(eng, name, z, T, weights[name], loss, dz[name], scale)
            else:
                ops.ntxent_fused(eng.dtype, z, T, weights[name], ws, loss, dz[name], dz_scale=scale)
        if z_meta is not None:
            # metadata terms: [cross_proj[m](first view of modality m) ; meta_proj(metadata)], positives = same sample
            Bm = z_meta.shape[0]
            zc = zs["cross0"]
            dmeta = torch.zeros(Bm, z_meta.shape[1], dtype=eng.tdt, device=dev)
            for half in (0, 1):
                zz = torch.cat([zc[half * Bm:(half + 1) * Bm], z_meta], 0)
                dd = torch.empty(2 * Bm, zz.shape[1], dtype=eng.tdt, device=dev)
                ws = eng._work("ntxent_ws", ops.ntxent_workspace_floats(2 * Bm, zz.shape[1]))
                if self.global_negatives:
                    self._ntxent_global(eng, "meta", zz, T, 0.5, loss, dd, sc["scale"] if sc is not None else None)
                else:
                    ops.ntxent_fused(eng.dtype, zz, T, 0.5, ws, loss, dd, dz_scale=sc["scale"] if sc is not None else None)
                dz["cross0"][half * Bm:(half + 1) * Bm] += dd[:Bm]
                dmeta += dd[Bm:]
            dz["meta"] = dmeta
        self._handles, self._handle_ranges = [], []
        # Single rank, no loss scaling: AdamW runs bucket by bucket as the gradients become final, on the lane that finished
        # them -- the other lane's backward hides it (one launch over all 81.65 M parameters at the end of the step is 0.45 ms
        # during which nothing else runs).  Data parallel: the buckets are all-reduced instead and AdamW follows the last
        # one; fp16: the overflow check needs every gradient first.
        early = sc is None and not self.dp and os.environ.get("SM3_ADAMW_BUCKETS", "1") != "0"
        # The bucket schedule (which gradient-ready notifications the backward pass emits) is a function of the engine, the
        # style and the branches in use -- static.  It is verified ONCE per such configuration, on a step that still runs
        # AdamW in one launch at the end: the notified ranges must tile [0, total) exactly, and a schedule that does not
        # raises BEFORE any parameter has been touched (never after half of the in-place updates).
        sched_key = (id(eng), self.style, metadata is not None, self.target_momentum is not None, st.total)
        verify = early and self.__dict__.get("_sched_ok") != sched_key
        if verify:
            early = False
            ranges = []
            eng.grad_ready = lambda f, l: ranges.append(self._bucket_range(eng, f, l))
        elif early:
            self.step_count += 1
            eng.grad_ready = lambda f, l: self._bucket_adamw(eng, f, l)
        else:
            eng.grad_ready = (lambda f, l: self._bucket_ready(eng, f, l)) if self.dp else None
        eng.backward(saved, dz)
        eng.grad_ready = None
        # Data parallel without loss scaling: AdamW follows every bucket's all-reduce on its own (the stream waits for
        # collective k, updates bucket k, while collectives k+1 ... are still on the wire) instead of one launch over all
        # 81.65 M parameters behind the LAST collective -- provided the notified ranges tile the buffer exactly
        # (they are the schedule the single-rank path verifies; anything else falls back to the single launch).
        dp_buckets = False
        # Peer-to-peer statistics exchange (opt-in): a timed-out exchange poisons its sums with NaN, which reach this rank's
        # gradients directly and every other rank's through the gradient all-reduce.  The update must not be applied then
        # (a caller that catches the error, or checkpoints on it, would hold a destroyed model): AdamW runs once,
        # behind the last collective, gated on "this rank's exchange failed OR any gradient is non-finite".
        p2p = self.__dict__.get("_p2p")
        if self.dp and sc is None and p2p is None and self._handles and len(self._handles) == len(self._handle_ranges) \
                and os.environ.get("SM3_ADAMW_BUCKETS", "1") != "0":
            pos = 0
            for a, b in sorted(self._handle_ranges):
                if a != pos or b <= a:
                    break
                pos = b
            dp_buckets = pos == st.total
        if dp_buckets:
            self.step_count += 1
            for h, (a, b) in zip(self._handles, self._handle_ranges):
                h.wait()
                ops.adamw(st.flat_p[a:b], st.flat_g[a:b], self.m[a:b], self.v[a:b], self.lr, self.betas[0], self.betas[1],
                          self.eps, self.wd, self.step_count, 1.0 / self.world)
        else:
            for h in self._handles:
                h.wait()
        if verify:
            pos = 0
            for a, b in sorted(ranges):
                if a != pos or b <= a:
                    break
                pos = b
            if pos != st.total or len(ranges) == 0:
                raise RuntimeError(f"gradient-ready notifications do not tile the {st.total} parameters exactly once "
                                   f"(contiguous cover ends at {pos}, {len(ranges)} notifications)")
            self._sched_ok = sched_key
        if early or dp_buckets:
            pass  # every bucket has had its AdamW launch (on the lane that finished it / behind its all-reduce)
        elif sc is None:
            skip = None
            if p2p is not None:
                skip = p2p.err.clone()
                ops.check_finite(st.flat_g, skip)
                if self.__dict__.get("_p2p_taken") is None:
                    self._p2p_taken = torch.full((1,), self.step_count, dtype=torch.int32, device=dev)
            self.step_count += 1
            ops.adamw(st.flat_p, st.flat_g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                      self.step_count, 1.0 / self.world, skip)
            if skip is not None:
                # optimizer steps APPLIED, counted where the decision was made (the error is sticky: once a step is skipped
                # every later one is, so step_count - 1 == taken whenever an update is applied)
                self._p2p_taken += (skip == 0).to(torch.int32)
                self._p2p_skip = skip
        else:
            # GradScaler.step() + update() (backbone_train.py:126-127) without a host synchronisation: inf / nan check
            # of the (all-reduced, still scaled) gradients, unscale + AdamW skipped on overflow, scale update
            ops.check_finite(st.flat_g, sc["found_inf"])
            ops.adamw_dynamic(st.flat_p, st.flat_g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                              self.wd, 1.0 / self.world, sc["scale"], sc["steps"], sc["found_inf"])
            ops.loss_scale_update(sc["scale"], sc["found_inf"], sc["tracker"], sc["steps"], self.scaler_cfg[1],
                                  self.scaler_cfg[2], self.scaler_cfg[3])
            self.step_count += 1  # calls made; the number of optimizer steps TAKEN lives on the device (steps_taken())
        if self.target_momentum is not None:
            skip = self.__dict__.pop("_p2p_skip", None)
            if skip is not None:
                # a skipped step must not move the target either: the EMA factor becomes 0 on the device, no host read
                w = (1.0 - self.target_momentum) * (skip == 0).to(torch.float32)
                self.flat_target.add_((st.flat_p - self.flat_target) * w)
            else:
                ops.ema_update(self.flat_target, st.flat_p, self.target_momentum)
        self.__dict__.pop("_p2p_skip", None)
        eng.weights_dirty = True  # raw-pointer writes: the engine's filter banks are stale
        self.loss = loss
        return loss

    def steps_taken(self):
        """Optimizer steps actually applied (a step skipped for an fp16 overflow does not count); synchronises."""
        if self._scaler is not None:
            return int(self._scaler["steps"])
        taken = self.__dict__.get("_p2p_taken")
        return int(taken.item()) if taken is not None else self.step_count

    # ---- checkpoint wire format: tools/backbone_train.py:578-587 ---------------------------
    def optimizer_state_dict(self):
        """torch.optim.AdamW-compatible state_dict (exp_avg / exp_avg_sq per parameter, shared step)."""
        eng = self._engine()
        eng.prepare(next(self.model.parameters()).device)
        st = eng.store
        state = {}
        if self.m is not None:  # before the first step torch.optim.AdamW's state is empty too
            taken = float(self.steps_taken())  # one host read, not one per parameter
            for i, n in enumerate(st.names):
                state[i] = {"step": torch.tensor(taken),
                            "exp_avg": st._view(self.m, n).clone(), "exp_avg_sq": st._view(self.v, n).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(st.names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        eng = self._engine()
        dev = next(self.model.parameters()).device
        eng.prepare(dev)
        st = eng.store
        self.m = torch.zeros(st.total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(st.total, dtype=torch.float32, device=dev)
        for i, n in enumerate(st.names):
            s = sd["state"].get(i)
            if s is None:
                continue
            st._view(self.m, n).copy_(s["exp_avg"])
            st._view(self.v, n).copy_(s["exp_avg_sq"])
            self.step_count = int(s["step"])
        if self._scaler is not None:
            self._scaler["steps"].fill_(self.step_count)
        g = sd["param_groups"][0]
        self.lr, self.wd, self.eps, self.betas = g["lr"], g["weight_decay"], g["eps"], tuple(g["betas"])
