"""Independent numpy re-derivation of the forward path, used ONLY to cross-check oracle/ (the C restatement).

Written from the public format definitions (AutoAWQ GEMM packing, AutoGPTQ v1, GGML block layouts, HF Llama),
deliberately sharing no code with oracle/*.c: dequantises whole matrices with vectorised numpy and runs plain
matmuls, so an indexing or nibble-order mistake in either implementation shows up as a mismatch.
"""
import numpy as np

AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]  # nibble position of column j inside a word: shift = 4*AWQ_ORDER[j]
# (/root/reference/src/loader/safetensors/awq.rs:29-32: shifts [0,16,4,20,8,24,12,28])


def round_act(x, act):
    x = np.asarray(x, dtype=np.float32)
    if act == "f16":
        return x.astype(np.float16).astype(np.float32)
    if act == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32).reshape(x.shape)
    return x


def awq_dequant(spec):
    K, N, gs = spec["K"], spec["N"], spec["group_size"]
    shifts = np.array([4 * p for p in AWQ_ORDER], dtype=np.uint32)
    qw = spec["qweight"].astype(np.uint32)
    q = ((qw[:, :, None] >> shifts[None, None, :]) & 0xF).reshape(K, N).astype(np.float32)
    z = ((spec["qzeros"].astype(np.uint32)[:, :, None] >> shifts[None, None, :]) & 0xF).reshape(K // gs, N)
    s = spec["scales"].astype(np.float32)
    g = np.arange(K) // gs
    return ((q - z[g].astype(np.float32)) * s[g]).T.copy()  # [N, K]


def gptq_dequant(spec):
    K, N, gs = spec["K"], spec["N"], spec["group_size"]
    sh = (4 * np.arange(8, dtype=np.uint32))
    qw = spec["qweight"].astype(np.uint32)  # [K/8, N]
    q = ((qw[:, None, :] >> sh[None, :, None]) & 0xF).reshape(K, N).astype(np.float32)
    z = ((spec["qzeros"].astype(np.uint32)[:, :, None] >> sh[None, None, :]) & 0xF).reshape(K // gs, N) + 1
    s = spec["scales"].astype(np.float32)
    g = spec["g_idx"] if spec.get("g_idx") is not None else np.arange(K) // gs
    return ((q - z[g].astype(np.float32)) * s[g]).T.copy()


def _f16(b):
    return np.ascontiguousarray(b).view(np.float16).astype(np.float32)


def q8_0_dequant(blocks, N, K):
    b = blocks.reshape(N, K // 32, 34)
    d = _f16(b[:, :, 0:2]).reshape(N, K // 32, 1)
    q = b[:, :, 2:].view(np.int8).astype(np.float32)
    return (d * q).reshape(N, K)


def q4_k_dequant(blocks, N, K):
    nb = K // 256
    b = blocks.reshape(N, nb, 144)
    d = _f16(b[:, :, 0:2]).reshape(N, nb, 1)
    dmin = _f16(b[:, :, 2:4]).reshape(N, nb, 1)
    sc = b[:, :, 4:16].astype(np.uint32)
    scales = np.zeros((N, nb, 8), dtype=np.uint32)
    mins = np.zeros((N, nb, 8), dtype=np.uint32)
    for j in range(8):
        if j < 4:
            scales[:, :, j] = sc[:, :, j] & 63
            mins[:, :, j] = sc[:, :, j + 4] & 63
        else:
            scales[:, :, j] = (sc[:, :, j + 4] & 0xF) | ((sc[:, :, j - 4] >> 6) << 4)
            mins[:, :, j] = (sc[:, :, j + 4] >> 4) | ((sc[:, :, j] >> 6) << 4)
    qs = b[:, :, 16:].reshape(N, nb, 4, 32)
    lo = (qs & 0xF).astype(np.float32)
    hi = (qs >> 4).astype(np.float32)
    q = np.stack([lo, hi], axis=3).reshape(N, nb, 8, 32)  # sub-block 2i = low nibbles, 2i+1 = high nibbles
    w = (d * scales.astype(np.float32))[..., None] * q - (dmin * mins.astype(np.float32))[..., None]
    return w.reshape(N, K)


def q6_k_dequant(blocks, N, K):
    nb = K // 256
    b = blocks.reshape(N, nb, 210)
    ql = b[:, :, 0:128].reshape(N, nb, 2, 64).astype(np.int32)
    qh = b[:, :, 128:192].reshape(N, nb, 2, 32).astype(np.int32)
    sc = b[:, :, 192:208].view(np.int8).reshape(N, nb, 2, 8).astype(np.float32)
    d = _f16(b[:, :, 208:210]).reshape(N, nb, 1, 1)
    out = np.zeros((N, nb, 2, 128), dtype=np.float32)
    l = np.arange(32)
    isx = l // 16
    q1 = ((ql[..., 0:32] & 0xF) | (((qh >> 0) & 3) << 4)) - 32
    q2 = ((ql[..., 32:64] & 0xF) | (((qh >> 2) & 3) << 4)) - 32
    q3 = ((ql[..., 0:32] >> 4) | (((qh >> 4) & 3) << 4)) - 32
    q4 = ((ql[..., 32:64] >> 4) | (((qh >> 6) & 3) << 4)) - 32
    out[..., 0:32] = d * sc[..., isx + 0] * q1
    out[..., 32:64] = d * sc[..., isx + 2] * q2
    out[..., 64:96] = d * sc[..., isx + 4] * q3
    out[..., 96:128] = d * sc[..., isx + 6] * q4
    return out.reshape(N, K)


def dequant(spec):
    k = spec["kind"]
    if k == "awq":
        return awq_dequant(spec)
    if k == "gptq":
        return gptq_dequant(spec)
    if k == "gguf":
        f = {8: q8_0_dequant, 12: q4_k_dequant, 14: q6_k_dequant}[spec["ggml_type"]]
        return f(spec["blocks"], spec["N"], spec["K"])
    w = spec["weight"]
    if w.dtype == np.uint16:
        return (w.astype(np.uint32) << 16).view(np.float32)
    return w.astype(np.float32)


def linear(spec, x):
    y = x.astype(np.float32) @ dequant(spec).T
    if spec.get("bias") is not None:
        y = y + spec["bias"].astype(np.float32)
    return y.astype(np.float32)


def rope_tables(cfg):
    hd, P = cfg["head_dim"], cfg["max_seq_len"]
    inv = 1.0 / (np.float64(cfg["rope_theta"]) ** (np.arange(0, hd, 2, dtype=np.float64) / hd))
    rs = cfg.get("rope_scaling")
    if rs and rs["type"] == "linear":
        inv = inv / rs["factor"]
    elif rs and rs["type"] == "llama3":
        f, lo, hi, old = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"]
        wl = 2 * np.pi / inv
        smooth = (old / wl - lo) / (hi - lo)
        mid = (1 - smooth) * inv / f + smooth * inv
        inv = np.where(wl > old / lo, inv / f, np.where(wl < old / hi, inv, mid))
    ang = np.arange(P, dtype=np.float32)[:, None] * inv.astype(np.float32)[None, :]
    return np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)


def rope(v, c, s, interleaved):
    half = v.shape[-1] // 2
    out = np.empty_like(v)
    if interleaved:
        x0, x1 = v[..., 0::2], v[..., 1::2]
        out[..., 0::2] = x0 * c - x1 * s
        out[..., 1::2] = x1 * c + x0 * s
    else:
        x0, x1 = v[..., :half], v[..., half:]
        out[..., :half] = x0 * c - x1 * s
        out[..., half:] = x1 * c + x0 * s
    return out


def rms_norm(x, w, eps, act):
    rs = 1.0 / np.sqrt(np.mean(x.astype(np.float32) ** 2, axis=-1, keepdims=True, dtype=np.float32) + np.float32(eps))
    return round_act(w * round_act(x * rs.astype(np.float32), act), act)


class NpLlama:
    """Token-at-a-time numpy forward with a python-list KV cache (small models only)."""

    def __init__(self, model):
        self.m = model
        self.cfg = model["config"]
        self.cos, self.sin = rope_tables(self.cfg)
        self.W = [{k: dequant(lay[k]) for k in ("q", "k", "v", "o", "gate", "up", "down")} for lay in model["layers"]]
        self.B = [{k: lay[k].get("bias") for k in ("q", "k", "v", "o", "gate", "up", "down")} for lay in model["layers"]]
        self.lm = dequant(model["lm_head"])
        self.emb = dequant(dict(kind="dense", weight=model["embed"]))
        self.K = [[] for _ in model["layers"]]
        self.V = [[] for _ in model["layers"]]

    def _lin(self, l, name, x):
        y = x @ self.W[l][name].T
        b = self.B[l][name]
        return (y + b.astype(np.float32)) if b is not None else y

    def step(self, token, pos):
        c = self.cfg
        act = c["act_dtype"]
        R = lambda a: round_act(a, act)
        nq, nkv, hd = c["n_heads"], c["n_kv_heads"], c["head_dim"]
        h = R(self.emb[token])
        prev = None
        for l, lay in enumerate(self.m["layers"]):
            if prev is not None:
                h = R(h + prev)
            xn = rms_norm(h, lay["attn_norm"], c["rms_eps"], act)
            q = R(self._lin(l, "q", xn)).reshape(nq, hd)
            k = R(self._lin(l, "k", xn)).reshape(nkv, hd)
            v = R(self._lin(l, "v", xn)).reshape(nkv, hd)
            q = R(rope(q, self.cos[pos], self.sin[pos], c["rope_interleaved"]))
            k = R(rope(k, self.cos[pos], self.sin[pos], c["rope_interleaved"]))
            self.K[l].append(k)
            self.V[l].append(v)
            Kc = np.stack(self.K[l], axis=1)  # [nkv, T, hd]
            Vc = np.stack(self.V[l], axis=1)
            rep = nq // nkv
            o = np.empty((nq, hd), dtype=np.float32)
            for hh in range(nq):
                s = (Kc[hh // rep] @ q[hh]) * np.float32(1.0 / np.sqrt(hd))
                p = np.exp(s - s.max())
                o[hh] = (p / p.sum()) @ Vc[hh // rep]
            o = R(o.reshape(-1))
            h = R(h + R(self._lin(l, "o", o)))
            xn = rms_norm(h, lay["ffn_norm"], c["rms_eps"], act)
            g = R(self._lin(l, "gate", xn))
            u = R(self._lin(l, "up", xn))
            a = R(R(g / (1.0 + np.exp(-g))) * u)
            prev = R(self._lin(l, "down", a))
        h = R(h + prev)
        xn = rms_norm(h, self.m["final_norm"], c["rms_eps"], act)
        return R(xn @ self.lm.T)


class NpLlamaTruth:
    """The same model with NO rounding anywhere: every tensor and every sum in float64 (weights are the exact dequantised values).
    This is what both the oracle and the HIP path approximate; tests/test_gpu_parity_truth.py measures each one's distance to it."""

    def __init__(self, model):
        self.m = model
        self.cfg = model["config"]
        c, s = rope_tables(self.cfg)
        self.cos, self.sin = c.astype(np.float64), s.astype(np.float64)
        f = lambda spec: dequant(spec).astype(np.float64)
        self.W = [{k: f(lay[k]) for k in ("q", "k", "v", "o", "gate", "up", "down")} for lay in model["layers"]]
        self.B = [{k: lay[k].get("bias") for k in ("q", "k", "v", "o", "gate", "up", "down")} for lay in model["layers"]]
        self.lm = f(model["lm_head"])
        self.emb = dequant(dict(kind="dense", weight=model["embed"]))
        self.K = [[] for _ in model["layers"]]
        self.V = [[] for _ in model["layers"]]

    def _lin(self, l, name, x):
        y = self.W[l][name] @ x
        b = self.B[l][name]
        return (y + b.astype(np.float64)) if b is not None else y

    @staticmethod
    def _norm(x, w, eps):
        return w.astype(np.float64) * (x / np.sqrt(np.mean(x * x) + eps))

    def step(self, token, pos):
        c = self.cfg
        nq, nkv, hd = c["n_heads"], c["n_kv_heads"], c["head_dim"]
        h = self.emb[token].astype(np.float64)
        for l, lay in enumerate(self.m["layers"]):
            xn = self._norm(h, lay["attn_norm"], c["rms_eps"])
            q = rope(self._lin(l, "q", xn).reshape(nq, hd), self.cos[pos], self.sin[pos], c["rope_interleaved"])
            k = rope(self._lin(l, "k", xn).reshape(nkv, hd), self.cos[pos], self.sin[pos], c["rope_interleaved"])
            v = self._lin(l, "v", xn).reshape(nkv, hd)
            self.K[l].append(k)
            self.V[l].append(v)
            Kc, Vc = np.stack(self.K[l], axis=1), np.stack(self.V[l], axis=1)
            rep = nq // nkv
            o = np.empty((nq, hd), dtype=np.float64)
            for hh in range(nq):
                s = (Kc[hh // rep] @ q[hh]) / np.sqrt(hd)
                p = np.exp(s - s.max())
                o[hh] = (p / p.sum()) @ Vc[hh // rep]
            h = h + self._lin(l, "o", o.reshape(-1))
            xn = self._norm(h, lay["ffn_norm"], c["rms_eps"])
            g, u = self._lin(l, "gate", xn), self._lin(l, "up", xn)
            h = h + self._lin(l, "down", (g / (1.0 + np.exp(-g))) * u)
        return self.lm @ self._norm(h, self.m["final_norm"], c["rms_eps"])


class NpMamba2:
    """Independent numpy Mamba2 step (vectorised over heads / state; public Mamba2 recurrence)."""

    def __init__(self, model):
        self.m, self.cfg = model, model["config"]
        c = self.cfg
        self.Win = [dequant(l["in_proj"]) for l in model["layers"]]
        self.Wout = [dequant(l["out_proj"]) for l in model["layers"]]
        self.lm = dequant(model["lm_head"])
        self.emb = dequant(dict(kind="dense", weight=model["embed"]))
        conv_dim = c["d_inner"] + 2 * c["n_groups"] * c["d_state"]
        self.conv = [np.zeros((conv_dim, c["conv_kernel"] - 1), np.float32) for _ in model["layers"]]
        self.ssm = [np.zeros((c["n_heads"], c["head_dim"], c["d_state"]), np.float32) for _ in model["layers"]]

    def step(self, token):
        c = self.cfg
        act = c["act_dtype"]
        R = lambda a: round_act(a, act)
        DI, NH, HD, NS, G = c["d_inner"], c["n_heads"], c["head_dim"], c["d_state"], c["n_groups"]
        conv_dim = DI + 2 * G * NS
        silu = lambda a: a / (1.0 + np.exp(-a))
        h = R(self.emb[token])
        for l, lay in enumerate(self.m["layers"]):
            xn = rms_norm(h, lay["norm"], c["rms_eps"], act)
            zx = R(xn @ self.Win[l].T)
            z, xraw, dtr = zx[:DI], zx[DI:DI + conv_dim], zx[DI + conv_dim:]
            win = np.concatenate([self.conv[l], xraw[:, None]], axis=1)
            a = R((win * lay["conv_w"]).sum(axis=1, dtype=np.float32) + lay["conv_b"])
            xbc = R(silu(a))
            self.conv[l] = win[:, 1:].copy()
            x = xbc[:DI].reshape(NH, HD)
            B = xbc[DI:DI + G * NS].reshape(G, NS)
            Cm = xbc[DI + G * NS:].reshape(G, NS)
            sp = R(dtr + lay["dt_bias"])
            dt = R(np.where(sp > 20, sp, np.log1p(np.exp(sp))))
            dA = np.exp(dt * -np.exp(lay["A_log"])).astype(np.float32)
            grp = np.arange(NH) // (NH // G)
            st = R(self.ssm[l] * dA[:, None, None] + (dt[:, None] * x)[:, :, None] * B[grp][:, None, :])
            self.ssm[l] = st
            y = R((st * Cm[grp][:, None, :]).sum(axis=2, dtype=np.float32) + lay["D"][:, None] * x).reshape(-1)
            y = R(y * R(silu(z)))
            yg = y.reshape(G, -1)
            rs = 1.0 / np.sqrt((yg.astype(np.float64) ** 2).mean(axis=1, keepdims=True).astype(np.float32) + np.float32(c["rms_eps"]))
            y = R(lay["gnorm"] * R(yg * rs).reshape(-1))
            h = R(h + R(y @ self.Wout[l].T))
        xn = rms_norm(h, self.m["final_norm"], c["rms_eps"], act)
        return R(xn @ self.lm.T)


class NpDsv2:
    """Independent numpy DeepSeek-V2 step in the NAIVE (HF modeling_deepseek.py) form: the cached latents are expanded to per-head
    k_nope / v through kv_b_proj for every cached token and ordinary attention is run -- no weight absorption.  In f32 this must
    agree with the oracle's absorbed form to float precision (it validates the algebra, the RoPE convention, the router and the
    MoE combine); in bf16 the two forms round at different places and agree only to a few bf16 ulps."""

    def __init__(self, model, truth=False):
        """truth=True: no rounding anywhere, every tensor and sum in float64 (what the oracle and the HIP path both approximate)"""
        self.m, self.cfg, self.truth = model, model["config"], truth
        dq = (lambda spec: dequant(spec).astype(np.float64)) if truth else dequant
        self.emb = dq(dict(kind="dense", weight=model["embed"]))
        self.lm = dq(model["lm_head"])
        self.W = []
        for lay in model["layers"]:
            w = {k: dq(lay[k]) for k in ("q_proj", "kv_a", "kv_b", "o")}
            if lay["is_moe"]:
                w["router"] = dq(lay["router"])
                w["experts"] = [{k: dq(e[k]) for k in ("gate", "up", "down")} for e in lay["experts"]]
                if "shared" in lay:
                    w["shared"] = {k: dq(lay["shared"][k]) for k in ("gate", "up", "down")}
            else:
                w.update({k: dq(lay[k]) for k in ("gate", "up", "down")})
            self.W.append(w)
        self.lat = [[] for _ in model["layers"]]

    def _rope(self, v, pos):
        c = self.cfg
        half = c["rope_dim"] // 2
        inv = 1.0 / (np.float64(c["rope_theta"]) ** (np.arange(half, dtype=np.float64) * 2 / c["rope_dim"]))
        ang = pos * inv
        cs, sn = (np.cos(ang), np.sin(ang)) if self.truth else (np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32))
        x = v.reshape(-1, half, 2)
        out = np.empty_like(x)
        out[..., 0] = x[..., 0] * cs - x[..., 1] * sn
        out[..., 1] = x[..., 1] * cs + x[..., 0] * sn
        return out.reshape(v.shape)

    def step(self, token, pos):
        c = self.cfg
        act = c["act_dtype"]
        truth = self.truth
        R = (lambda a: a) if truth else (lambda a: round_act(a, act))
        fdt = np.float64 if truth else np.float32
        rms_norm_ = (lambda x, w, eps, _a: np.asarray(w, np.float64) * (x / np.sqrt(np.mean(x * x) + eps))) if truth else rms_norm
        NH, RK, DN, DR, DV = c["n_heads"], c["kv_lora_rank"], c["nope_dim"], c["rope_dim"], c["v_dim"]
        silu = lambda a: a / (1.0 + np.exp(-a))
        mlp = lambda w, x: R(R(R(silu(R(x @ w["gate"].T))) * R(x @ w["up"].T)) @ w["down"].T)
        h = R(self.emb[token])
        for l, lay in enumerate(self.m["layers"]):
            w = self.W[l]
            xn = rms_norm_(h, lay["attn_norm"], c["rms_eps"], act)
            q = R(xn @ w["q_proj"].T).reshape(NH, DN + DR)
            kva = R(xn @ w["kv_a"].T)
            lat = rms_norm_(kva[:RK], lay["kv_norm"], c["rms_eps"], act)
            kpe = R(self._rope(kva[RK:], pos))
            self.lat[l].append(np.concatenate([lat, kpe]))
            Cm = np.stack(self.lat[l])                                   # [T, RK + DR]
            kv = R(Cm[:, :RK] @ w["kv_b"].T).reshape(-1, NH, DN + DV)      # naive expansion
            qpe = R(self._rope(q[:, DN:], pos))
            sc = (np.einsum("hd,thd->ht", q[:, :DN], kv[:, :, :DN]) + qpe @ Cm[:, RK:].T) / np.sqrt(fdt(DN + DR))
            p = np.exp(sc - sc.max(axis=1, keepdims=True))
            p = p / p.sum(axis=1, keepdims=True)
            att = R(np.einsum("ht,thd->hd", p, kv[:, :, DN:])).reshape(-1)
            h = R(h + R(att @ w["o"].T))
            xn = rms_norm_(h, lay["ffn_norm"], c["rms_eps"], act)
            if not lay["is_moe"]:
                out = mlp(w, xn)
            else:
                lg = (xn @ w["router"].T).astype(fdt)
                s = np.exp(lg - lg.max())
                s = s / s.sum()
                sel = np.argsort(-s, kind="stable")[:c["top_k"]]
                wt = s[sel] / (s[sel].sum() + 1e-20) * c["routed_scale"] if c["norm_topk"] else s[sel] * c["routed_scale"]
                routed = np.zeros_like(h)
                for e, we in zip(sel, wt):
                    routed = routed + fdt(we) * mlp(w["experts"][int(e)], xn)
                out = R(routed)
                if "shared" in w:
                    out = R(out + mlp(w["shared"], xn))
            h = R(h + out)
        xn = rms_norm_(h, self.m["final_norm"], c["rms_eps"], act)
        return R(xn @ self.lm.T)
