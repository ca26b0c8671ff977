"""Differential fuzz of the attention entry points (crab_attn_fwd: 64- and 128-row kernels, head_dim 32 / 64 / 128, causal, GQA, gated bias, left-pad
kv_start, general key_mask; crab_attn_decode / _masked / _keymask incl. the grouped-query kernel and device-resident context lengths) against
fp32 torch attention on the same bf16 operands.   python scripts/fuzz_attn.py [cases] [seed]"""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops, _lib

BF = torch.bfloat16
NCASE = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, rejected, done, why = [], 0, 0, {}


def reject(e):
    global rejected
    msg = str(e)
    if "error -1:" in msg or "error -3:" in msg:
        rejected += 1
        k = msg.split(":", 2)[-1].strip()[:90]
        why[k] = why.get(k, 0) + 1
        return True
    return False


for case in range(NCASE):
    g = torch.Generator(device="cuda").manual_seed(1000 + case)
    d = rng.choice([32, 64, 64, 128, 128])
    Hk = rng.choice([1, 2, 4])
    G = rng.choice([1, 1, 2, 4, 7])
    H = Hk * G
    B = rng.choice([1, 2, 3, 5])
    decode = rng.random() < 0.4 and d != 32
    if not decode:
        Sq = rng.choice([1, 7, 32, 63, 64, 65, 127, 128, 129, 200, 257, 300, 702])
        causal = rng.random() < 0.6 and d != 32
        Skv = Sq if causal or rng.random() < 0.5 else rng.choice([Sq, 48, 256, 600])
        use_bias = (not causal) and d != 32 and rng.random() < 0.25
        mode = rng.choice(["none", "none", "kv_start", "key_mask"]) if d != 32 and not use_bias else "none"
        q = (torch.randn(B, Sq, H, d, device="cuda", generator=g) * 0.7).to(BF)
        k = (torch.randn(B, Hk, Skv, d, device="cuda", generator=g) * 0.7).to(BF)
        v = (torch.randn(B, Hk, Skv, d, device="cuda", generator=g) * 0.7).to(BF)
        Sp = (Skv + 7) // 8 * 8
        vt = torch.zeros(B, Hk, d, Sp, device="cuda", dtype=BF); vt[..., :Skv] = v.transpose(2, 3)
        obuf = torch.full((B * Sq + 2, H * d), 777.0, device="cuda", dtype=BF)          # guard rows above and below the output
        o = obuf[1:B * Sq + 1].view(B, Sq, H * d)
        scale = 1.0 / math.sqrt(d)
        keyok = torch.ones(B, Skv, device="cuda", dtype=torch.bool)
        kw = {}
        if mode == "kv_start":
            ks = torch.tensor([rng.randrange(0, max(1, Skv // 2)) for _ in range(B)], device="cuda", dtype=torch.int32)
            keyok = torch.arange(Skv, device="cuda")[None] >= ks[:, None]
            kw["kv_start"] = ks
        elif mode == "key_mask":
            keyok = torch.rand(B, Skv, device="cuda", generator=g) > 0.3
            kw["key_mask"] = ops.pack_key_mask(keyok)
        bias = gate = None
        if use_bias:
            bias = torch.randn(H, Sq, Skv, device="cuda", generator=g) * 0.5
            gate = torch.rand(B, H, Sq, device="cuda", generator=g) * 2
            kw["bias"], kw["gate"] = bias, gate
        desc = f"case {case}: fwd B={B} H={H} Hk={Hk} Sq={Sq} Skv={Skv} d={d} causal={causal} bias={use_bias} mask={mode}"
        try:
            ops.attn_fwd(q, k, vt, o, q_strides=(Sq * H * d, d, H * d), k_strides=(Hk * Skv * d, Skv * d, d), vt_strides=(Hk * d * Sp, d * Sp, Sp),
                         o_strides=(Sq * H * d, H * d), B=B, H=H, Hk=Hk, Sq=Sq, Skv=Skv, head_dim=d, scale=scale, causal=causal, **kw)
            torch.cuda.synchronize()
        except _lib.CrabHipError as e:
            if not reject(e): bad.append(desc + " -> " + str(e)[:200])
            continue
        kk = k.float().repeat_interleave(G, 1); vv = v.float().repeat_interleave(G, 1)
        sc = torch.einsum("bshd,bhtd->bhst", q.float(), kk) * scale
        if use_bias: sc = sc + gate[:, :, :, None] * bias[None]
        allow = keyok[:, None, None, :].expand(B, H, Sq, Skv).clone()
        if causal: allow &= (torch.arange(Skv, device="cuda")[None, :] <= torch.arange(Sq, device="cuda")[:, None] + (Skv - Sq))[None, None]
        sc = sc.masked_fill(~allow, float("-inf"))
        p = torch.softmax(sc, -1).nan_to_num(0.0)
        ref = torch.einsum("bhst,bhtd->bshd", p, vv).reshape(B, Sq, H * d)
        err = float((o.float() - ref).abs().max())
        done += 1
        if not (err < 2.5e-2) or not torch.isfinite(o.float()).all(): bad.append(desc + f" -> max abs err {err:.3e}")
        if not (bool((obuf[0] == 777.0).all()) and bool((obuf[-1] == 777.0).all())): bad.append(desc + " -> a store landed outside the output")
    else:
        Tmax = rng.choice([64, 128, 960])
        ctx = rng.randrange(1, Tmax + 1)
        Bd = rng.choice([1, 3, 8, 40, 130]) if Hk * 130 <= 600 else rng.choice([1, 3, 8])
        mode = rng.choice(["none", "none", "kv_start", "key_mask"])
        dev_ctx = rng.random() < 0.5
        kc = (torch.randn(Bd, Hk, Tmax, d, device="cuda", generator=g) * 0.7).to(BF)
        vc = (torch.randn(Bd, Hk, Tmax, d, device="cuda", generator=g) * 0.7).to(BF)
        q = (torch.randn(Bd, H * d, device="cuda", generator=g) * 0.7).to(BF)
        obuf = torch.full((Bd + 2, H * d), 777.0, device="cuda", dtype=BF)
        o = obuf[1:Bd + 1]
        scale = 1.0 / math.sqrt(d)
        keyok = torch.zeros(Bd, Tmax, device="cuda", dtype=torch.bool); keyok[:, :ctx] = True
        kw = {}
        if mode == "kv_start":
            ks = torch.tensor([rng.randrange(0, ctx) for _ in range(Bd)], device="cuda", dtype=torch.int32)
            keyok &= torch.arange(Tmax, device="cuda")[None] >= ks[:, None]
            kw["kv_start"] = ks
        elif mode == "key_mask":
            keyok &= torch.rand(Bd, Tmax, device="cuda", generator=g) > 0.3
            kw["key_mask"] = ops.pack_key_mask(keyok)
        desc = f"case {case}: decode B={Bd} H={H} Hk={Hk} d={d} Tmax={Tmax} ctx={ctx} dev_ctx={dev_ctx} mask={mode}"
        try:
            if dev_ctx:
                pos = torch.full((1,), ctx - 1, device="cuda", dtype=torch.int32)
                ops.attn_decode(q, kc, vc, o, Bd, H, Hk, d, Tmax, 1, scale, ctx_dev=pos, **kw)
            else:
                ops.attn_decode(q, kc, vc, o, Bd, H, Hk, d, Tmax, ctx, scale, **kw)
            torch.cuda.synchronize()
        except _lib.CrabHipError as e:
            if not reject(e): bad.append(desc + " -> " + str(e)[:200])
            continue
        kk = kc.float().repeat_interleave(G, 1); vv = vc.float().repeat_interleave(G, 1)
        sc = torch.einsum("bhd,bhtd->bht", q.float().view(Bd, H, d), kk) * scale
        sc = sc.masked_fill(~keyok[:, None, :], float("-inf"))
        p = torch.softmax(sc, -1).nan_to_num(0.0)
        ref = torch.einsum("bht,bhtd->bhd", p, vv).reshape(Bd, H * d)
        err = float((o.float() - ref).abs().max())
        done += 1
        if not (err < 2.5e-2) or not torch.isfinite(o.float()).all(): bad.append(desc + f" -> max abs err {err:.3e}")
        if not (bool((obuf[0] == 777.0).all()) and bool((obuf[-1] == 777.0).all())): bad.append(desc + " -> a store landed outside the output")
print(f"{done} cases computed, {rejected} rejected by the library, {len(bad)} failures")
for k_, v_ in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  rejected x{v_}: {k_}")
for b_ in bad[:40]: print("FAIL", b_)
sys.exit(1 if bad else 0)
