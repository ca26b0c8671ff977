"""Thin tensor-level wrappers over the C-ABI (plumbing only: pointers, strides, the current HIP stream).

Every function launches HIP kernels from libcrab_hip.so on torch's current stream and returns/fills
caller-visible torch tensors.  No arithmetic happens in PyTorch here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc

ACT = {"none": 0, None: 0, "gelu": 1, "quick_gelu": 2, "relu": 3, "silu": 4, "swiglu_pair": 5}
BF16 = torch.bfloat16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise _lib.CrabHipError("crab_amd ops need CUDA/HIP tensors (no CPU fallback exists)")
    return t.device.index or 0


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _chk_bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != BF16:
            raise _lib.CrabHipError(f"expected bfloat16 storage, got {t.dtype}")


class launch_trace:
    """`with ops.launch_trace(device) as t: ...` -> t.counts = {launch-site name: launches} of every kernel the C-ABI entry points issued inside the
    block (crab_trace_begin / crab_trace_end, include/crab_hip.h).  Names carry the template instantiation where the dispatch has one
    ("attn_decode_kernel<128>", "gemm_bt_ring_kernel+rope2").  The parity tests use it to prove which kernel a fixture was compared through."""

    def __init__(self, device: int = 0):
        self.device = device
        self.counts = {}

    def __enter__(self):
        _lib.check(_lib.load().crab_trace_begin(_lib.ctx(self.device)), self.device)
        return self

    def __exit__(self, *exc):
        lib, h = _lib.load(), _lib.ctx(self.device)
        need = lib.crab_trace_end(h, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        lib.crab_trace_end(h, buf, len(buf))
        self.counts = {}
        for line in buf.value.decode().splitlines():
            name, _, n = line.rpartition("\t")
            self.counts[name] = int(n)
        return False

    def launched(self, name: str) -> int:
        return self.counts.get(name, 0)


class KernelProfiler:
    """Optional HIP-event timing of kernel launches on the current stream (bench.py roofline leg).  Only used outside
    graph capture; adds two event records per profiled launch.  GEMM launches (M >= min_m) are bucketed by kernel
    variant with their algorithmic FLOPs; the decode-attention kernel is sampled with its algorithmic KV bytes on the
    decode steps GenerationEngine.generate runs eagerly (every `decode_every`-th step; the eager step is bit-identical
    to the HIP-graph replay it stands in for, tests/test_model_gpu.py)."""

    def __init__(self, min_m: int = 512, decode_every: int = 32, phase_only: bool = False):
        """phase_only: record nothing but the three phase marks of a generate() call (encode_begin / prefill_end / decode_end) - the SHIPPED path
        runs underneath (native layer sequencers, every decode step a graph replay); three event records per call are the whole footprint."""
        self.phase_only = phase_only
        self.min_m = (1 << 62) if phase_only else min_m
        self.decode_every = 0 if phase_only else decode_every
        self.decode_ctx = 0          # live KV length of the sampled step (host mirror of the device position word)
        self.decode_eager = False    # set by GenerationEngine.generate around a sampled eager decode step
        self.marks = []              # (phase name, event)
        self.records = []            # (variant, work, start_event, end_event); work = FLOPs ("gemm*") or bytes ("attn_decode*")

    def mark(self, name: str):
        """Phase boundary on the current stream (encode_begin / prefill_end / decode_end of one generate call)."""
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def phase_ms(self):
        """Sum over generate calls of (encoders + prefill) and decode time, from the phase marks."""
        torch.cuda.synchronize()
        pre = dec = 0.0
        last = {}
        for name, e in self.marks:
            if name == "prefill_end" and "encode_begin" in last:
                pre += last["encode_begin"].elapsed_time(e)
            if name == "decode_end" and "prefill_end" in last:
                dec += last["prefill_end"].elapsed_time(e)
            last[name] = e
        return pre, dec

    def summary(self):
        """{kernel variant: {launches, work, ms, classes: {class: {launches, work, ms}}}}; a record's variant may carry a class after
        '|' (GEMMs: 'decoder' = hyper-LoRA projections of the decoder, 'encoder' = CLIP / BEATs / Q-Former / head shapes)."""
        torch.cuda.synchronize()
        out = {}
        for variant, work, e0, e1 in self.records:
            variant, _, cls = variant.partition("|")
            ms = e0.elapsed_time(e1)
            d = out.setdefault(variant, {"launches": 0, "work": 0.0, "ms": 0.0, "classes": {}})
            for tgt in (d, d["classes"].setdefault(cls, {"launches": 0, "work": 0.0, "ms": 0.0}) if cls else None):
                if tgt is not None:
                    tgt["launches"] += 1
                    tgt["work"] += work
                    tgt["ms"] += ms
        return out


GemmProfiler = KernelProfiler

PROFILER: Optional[KernelProfiler] = None


def per_launch_profiling() -> bool:
    """True while a profiler that times individual launches is attached: the layer sequences then run launch by launch from Python (same launches,
    same order) so that ops.gemm sees every GEMM; a phase-only profiler leaves the native C sequencers in place."""
    return PROFILER is not None and not PROFILER.phase_only


def _variant(M: int, N: int, batch: int = 1, K: int = 4096) -> str:
    """Mirror of the tile choice in csrc/gemm.hip (crab_gemm_bf16) - the bucket NAME of the profiler's records."""
    if batch == 1 and 256 < M <= DECODE_MAX_ROWS:
        return "gemm_dec2_kernel"                     # two 256-row groups per block (a workspace is always handed in at these row counts)
    if batch == 1 and 64 < M <= 256:
        return "gemm_dec_ws_kernel"
    big = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if M <= 64 or N <= 64 or big < 192:
        return "gemm_bt_kernel<64,64>"
    big256 = ((M + 255) // 256) * ((N + 255) // 256) * batch
    rounds = (big256 + 255) // 256
    ring = big256 >= 120 and big256 * 100 >= rounds * 256 * 55 and M >= 1024 and N >= 1024 and K >= 1024
    return "gemm_bt_ring_kernel<256,256>" if ring else "gemm_bt_glds_kernel<128,128>"


# The residual stream (decoder x, CLIP tower x, the pre-LN sums of the post-LN encoders) is kept in fp32: the reference adds its residuals in
# fp32 (models/modeling_llama.py:805-827 run with --bf16 False, scripts/quick_start.sh:42-44) and a bf16 rounding of x per sub-layer is what put
# the 32-layer logits 1.7e-2 away from it.  CRAB_RESIDUAL_FP32=0 restores the all-bf16 storage of r01-r03 (A/B runs only).
RESIDUAL_FP32 = os.environ.get("CRAB_RESIDUAL_FP32", "1") != "0"
RES_DTYPE = torch.float32 if RESIDUAL_FP32 else torch.bfloat16

# The LayerNorm weights / biases of the encoders (CLIP, BEATs, both Q-Formers, the projector norms, SegModule) are kept in fp32 (r05): they are
# not matrix operands, and rounding them to bf16 put a systematic 2^-9 relative error on every channel of every LayerNorm output - the largest
# single removable contribution to the encoders' distance from the fp32 reference (scripts/parity_floor.py).  CRAB_NORM_FP32=0
# restores the bf16 parameters of r01-r04 (A/B runs only).  The decoder's RMSNorm weights stay bf16 (their row-owning fused forms are tuned at the
# instruction level; measured cost 6e-4 of the logit scale on the tiny stacks).
NORM_PARAMS_FP32 = os.environ.get("CRAB_NORM_FP32", "1") != "0"
NORM_DTYPE = torch.float32 if NORM_PARAMS_FP32 else torch.bfloat16
# ... and so are the decoder's RMSNorm weights (input_layernorm / post_attention_layernorm / model.norm) wherever the residual stream is fp32
# (crab_gemm_desc.norm_w_fp32, crab_llama_layer.norm_w_fp32, crab_rmsnorm_p): with a real checkpoint's weights their bf16 rounding is worth
# 6e-4 of the logit scale on the 2-layer fixtures (the synthetic full-size model has norm weights of exactly 1, which hides it).
RMS_DTYPE = torch.float32 if (NORM_PARAMS_FP32 and os.environ.get("CRAB_RESIDUAL_FP32", "1") != "0") else torch.bfloat16

# CRAB_DECODE_MAX_ROWS, read from include/crab_hip.h (one source; _lib.load() checks it against the built library): up to this many rows a
# GEMM with a workspace streams the weights once
DECODE_MAX_ROWS = _lib.header_const("CRAB_DECODE_MAX_ROWS")

ROWFIN = os.environ.get("CRAB_ROWFIN", "1") != "0"     # the M <= 16 layer tail of csrc/rowfin.hip (same switch, same parse as the library: off iff the value is exactly "0")


def rowfin_lora_ok(nl: int, r: int, N: int) -> bool:
    """crab_rowfin_lora_ok: the library's own predicate for the in-call hyper-LoRA form (lora_self) - the one run_group (csrc/llama_layer.hip) uses."""
    return bool(_lib.load().crab_rowfin_lora_ok(int(nl), int(r), int(N)))

_SPLITK_WS = {}
WS_SLOT = 0      # scratch slot of the launches being issued/captured: groups decoding concurrently on different HIP streams
                 # (GenerationEngine.generate, decode_streams > 1) must not share the split-K partial slabs


def _splitk_workspace(device) -> torch.Tensor:
    """One caller-owned split-K scratch per device (stable address: safe under HIP-graph capture)."""
    key = (device.index or 0, WS_SLOT)
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.empty((256 << 20,), device=device, dtype=torch.uint8)
    return _SPLITK_WS[key]


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none",
         residual: Optional[torch.Tensor] = None, res_scale: float = 1.0, x2: Optional[torch.Tensor] = None,
         w2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_fp32: bool = False, tune: int = 0,
         post_norm=None, rope=None, route=None, lora_self=None, info: Optional[dict] = None, prof_class: Optional[str] = None,
         rope_row_off: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = res_scale*residual + act(x[M,K] @ w[N,K]^T + x2 @ w2^T + bias).  2-D row-strided operands.
    rope = (tab, k_cache, v_cache, H, Hk, d, Tmax, pos0, pos_dev): packed q|k|v projection of ONE row per sequence followed
    by RoPE + KV-cache append (== qkv_rope_split(B=M, S=1) on out), fused into the split-K reduction when there is one.
    rope = (..., pos_dev = None, S, pos_ids[, vt]): the PREFILL form, S rows per sequence - when the library says so (info["fused_prefill_rope"]
    = crab_gemm_fuses_prefill_rope: 1 or 2) q is rotated in place and k rotated into the cache by the projection's epilogue; at 1 the caller
    finishes with qkv_rope_split(rope_tab=None, k_cache=None) for the v columns, at 2 (vt given) the epilogue did those too; at 0 the caller
    runs the full qkv_rope_split as before.
    route = (RA, nproj, nl, r, ucols, scaling, u_out) (with post_norm, M <= DECODE_MAX_ROWS): u_out = hyperlora_route(post-norm rows, RA)
    for the NEXT projection group, computed inside the row-owning reduction kernel when that path is taken.
    lora_self = (RA, nl, r, scaling, lora_B) (with post_norm, M <= 16, no x2): the hyper-LoRA update of THIS single-projection group is
    evaluated inside the call - its router rows ride on the projection's launch, the update is applied by the M <= 16 layer tail.
    rope_row_off (int32 [M], decode form of `rope` only): the RAGGED decode batch - row m is rotated at slot - rope_row_off[m] while its
    K / V rows land in the common slot (crab_gemm_desc.rope_row_off).
    act == "swiglu_pair": w rows are interleaved (gate_i, up_i) and out is [M, N/2] = silu(gate) * up."""
    _chk_bf16(x, w, bias, x2, w2)
    if residual is not None and residual.dtype not in (BF16, torch.float32):
        raise _lib.CrabHipError(f"residual must be bfloat16 or float32, got {residual.dtype}")
    d = _dev(x)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N // 2 if act == "swiglu_pair" else N), device=x.device, dtype=torch.float32 if out_fp32 else BF16)
    g = GemmDesc()
    g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.R = residual.data_ptr() if residual is not None else None
    g.lda, g.ldb, g.ldc = x.stride(0), w.stride(0), out.stride(0)
    g.ldr = residual.stride(0) if residual is not None else 0
    g.r_fp32 = 1 if (residual is not None and residual.dtype == torch.float32) else 0
    if x2 is not None:
        assert w2 is not None and x2.shape[1] == w2.shape[1] and x2.shape[0] == M and w2.shape[0] == N
        g.A2, g.B2, g.lda2, g.ldb2, g.K2 = x2.data_ptr(), w2.data_ptr(), x2.stride(0), w2.stride(0), x2.shape[1]
    g.M, g.N, g.K = M, N, K
    g.act = ACT[act]
    g.c_fp32 = 1 if out.dtype == torch.float32 else 0
    g.res_scale = res_scale
    g.batch, g.nb0 = 1, 1
    g.tune = tune
    if post_norm is not None:                      # (weight, eps, out): out = rmsnorm(result) * weight, fused when possible
        g.norm_w, g.norm_eps, g.norm_out, g.ld_norm = post_norm[0].data_ptr(), post_norm[1], post_norm[2].data_ptr(), post_norm[2].stride(0)
        g.norm_w_fp32 = 1 if post_norm[0].dtype == torch.float32 else 0
    if route is not None:
        assert post_norm is not None and M <= DECODE_MAX_ROWS
        rRA, rnp, rnl, rr, ruc, rsc, ru = route
        g.route_RA, g.route_U, g.route_ldra, g.route_ldu = rRA.data_ptr(), ru.data_ptr(), rRA.stride(0), ru.stride(0)
        g.route_nproj, g.route_nl, g.route_r, g.route_ucols, g.route_scaling = rnp, rnl, rr, ruc, rsc
    if lora_self is not None:
        assert x2 is None and post_norm is not None
        lRA, lnl, lr, lsc, lB = lora_self
        _chk_bf16(lRA, lB)
        g.lora_RA, g.lora_ldra, g.lora_nl, g.lora_r, g.lora_scaling = lRA.data_ptr(), lRA.stride(0), lnl, lr, lsc
        g.B2, g.ldb2, g.K2 = lB.data_ptr(), lB.stride(0), lB.shape[1]
    if rope is not None and len(rope) > 9 and int(rope[9]) <= 1:
        # the prefill form with ONE row per sequence: the library would read these fields as the decode form (rope_S <= 1: rotation at pos0 /
        # pos_dev, no rotary position ids) and still answer "not fused" to the prefill question - the caller's split pass would then rotate a
        # second time (harmless only at position 0).  Nothing to fuse at one row per sequence: plain projection, the caller runs the split.
        if info is not None:
            info["fused_prefill_rope"] = 0
        rope = None
    if rope is not None:
        tab, kcache, vcache, rH, rHk, rd, rT, rp0, rpd = rope[:9]
        g.rope_tab, g.rope_k_cache, g.rope_v_cache = tab.data_ptr(), kcache.data_ptr(), vcache.data_ptr()
        g.rope_pos_dev = rpd.data_ptr() if rpd is not None else None
        g.rope_H, g.rope_Hk, g.rope_d, g.rope_Tmax, g.rope_pos0 = rH, rHk, rd, rT, rp0
        if rope_row_off is not None:
            if len(rope) > 9:
                raise ValueError("rope_row_off belongs to the decode form of the fused RoPE")
            if rope_row_off.dtype != torch.int32 or rope_row_off.numel() < M or not rope_row_off.is_contiguous():
                raise ValueError("rope_row_off must be a contiguous int32 [M] tensor")
            g.rope_row_off = rope_row_off.data_ptr()
        if len(rope) > 9:                          # prefill form: S rows per sequence (+ optional rotary positions [B, S] int32)
            g.rope_S = int(rope[9])
            pid = rope[10] if len(rope) > 10 else None
            if pid is not None:
                assert pid.dtype == torch.int32 and pid.stride(-1) == 1
                g.rope_pos_ids, g.rope_ld_pos = pid.data_ptr(), pid.stride(0)
            if len(rope) > 11 and rope[11] is not None:           # V^T scratch [B, Hk, d, vt_ld]: the v columns in the epilogue too
                g.rope_vt, g.rope_vt_ld = rope[11].data_ptr(), rope[11].stride(-2)
            if info is not None:                                   # 0: not fused; 1: q / k; 2: q / k / v (no split pass left)
                info["fused_prefill_rope"] = int(_lib.load().crab_gemm_fuses_prefill_rope(C.byref(g)))
    if M <= DECODE_MAX_ROWS:
        ws = _splitk_workspace(x.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()
    prof = PROFILER
    if prof is not None and M >= prof.min_m and not torch.cuda.is_current_stream_capturing():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.load().crab_gemm_bf16(_lib.ctx(d), _stream(), C.byref(g)), d)
        e1.record()
        # class of the launch for bench.py's by_class split: stated by the caller (PackedLinearGroup.prof_class: "decoder" for every projection
        # of the decoder stack, adapted or not; "head" for lm_head); un-tagged launches are the encoders' / projectors'
        k2 = x2.shape[1] if x2 is not None else (lora_self[4].shape[1] if lora_self is not None else 0)
        prof.records.append((_variant(M, N, 1, K) + "|" + (prof_class or "encoder"), 2.0 * M * N * (K + k2), e0, e1))
        return out
    _lib.check(_lib.load().crab_gemm_bf16(_lib.ctx(d), _stream(), C.byref(g)), d)
    return out


def gemm_desc(desc: GemmDesc, device: int = 0):
    """Raw descriptor launch (batched / strided forms)."""
    _lib.check(_lib.load().crab_gemm_bf16(_lib.ctx(device), _stream(), C.byref(desc)), device)


def hyperlora_mix(t: torch.Tensor, nproj: int, nl: int, r: int, ucols: int, scaling: float,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = _dev(t)
    M = t.shape[0]
    if out is None:
        out = torch.empty((M, ucols), device=t.device, dtype=BF16)
    _lib.check(_lib.load().crab_hyperlora_mix(_lib.ctx(d), _stream(), _p(t), t.stride(0), 1 if t.dtype == torch.float32 else 0,
                                              _p(out), out.stride(0), M, nproj, nl, r, ucols, scaling), d)
    return out


def hyperlora_route_workspace(M: int, K: int, tcols: int) -> int:
    return int(_lib.load().crab_hyperlora_route_workspace(M, K, tcols))


def hyperlora_route(x: torch.Tensor, ra: torch.Tensor, nproj: int, nl: int, r: int, ucols: int, scaling: float,
                    out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """U[M,ucols] = routing mix of x @ [R;A]^T (split-K skinny product + fixed-order reduce + fp32 softmax)."""
    _chk_bf16(x, ra)
    d = _dev(x)
    M, K = x.shape
    if out is None:
        out = torch.empty((M, ucols), device=x.device, dtype=BF16)
    need = hyperlora_route_workspace(M, K, ra.shape[0])
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need,), device=x.device, dtype=torch.uint8)
    _lib.check(_lib.load().crab_hyperlora_route(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(ra), ra.stride(0), M, K, nproj, nl, r,
                                                _p(out), out.stride(0), ucols, scaling, _p(workspace),
                                                workspace.numel() * workspace.element_size()), d)
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16, or fp32 (a row of the fp32 residual stream: no intermediate rounding of x_hat); w bf16 or fp32 (RMS_DTYPE); out bf16."""
    d = _dev(x)
    M, D = x.shape
    if out is None:
        out = torch.empty((M, D), device=x.device, dtype=BF16)
    wf = w.dtype == torch.float32
    if not wf:
        _chk_bf16(w)
    if x.dtype != torch.float32:
        _chk_bf16(x)
    _lib.check(_lib.load().crab_rmsnorm_p(_lib.ctx(d), _stream(), _p(x), 1 if x.dtype == torch.float32 else 0, x.stride(0), _p(w), 1 if wf else 0,
                                          _p(out), out.stride(0), M, D, eps), d)
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 or fp32 rows (residual stream / pre-LN sums); w / b bf16 or - both - fp32 (NORM_DTYPE); out bf16."""
    d = _dev(x)
    M, D = x.shape
    if out is None:
        out = torch.empty((M, D), device=x.device, dtype=BF16)
    wf = w.dtype == torch.float32
    if not wf:
        _chk_bf16(w, b)
    elif b is not None and b.dtype != torch.float32:
        raise _lib.CrabHipError("layernorm: weight and bias must have the same storage (both bfloat16 or both float32)")
    if x.dtype != torch.float32:
        _chk_bf16(x)
    _lib.check(_lib.load().crab_layernorm_p(_lib.ctx(d), _stream(), _p(x), 1 if x.dtype == torch.float32 else 0, x.stride(0), _p(w), _p(b),
                                            1 if wf else 0, _p(out), out.stride(0), M, D, eps), d)
    return out


def embedding(ids: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_bf16(table)
    d = _dev(table)
    ids = ids.reshape(-1).to(device=table.device, dtype=torch.int64).contiguous()
    T, D = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((T, D), device=table.device, dtype=BF16)
    if T:
        fn = _lib.load().crab_embedding_f32 if out.dtype == torch.float32 else _lib.load().crab_embedding      # fp32: the residual stream's first row
        _lib.check(fn(_lib.ctx(d), _stream(), _p(ids), _p(table), _p(out), out.stride(0), T, D, table.shape[0]), d)
    return out


def cast_rows(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, lds: Optional[int] = None):
    """dst[r, :cols] = src[r, :cols] across bf16 <-> fp32 (or a plain copy when the dtypes agree); row-strided 2-D views."""
    d = _dev(src)
    lds = src.stride(0) if lds is None else lds
    if src.dtype == dst.dtype:
        return copy_rows(src, dst, rows, cols, lds=lds)
    if src.dtype == BF16 and dst.dtype == torch.float32:
        _lib.check(_lib.load().crab_cast_rows_bf16_f32(_lib.ctx(d), _stream(), _p(src), lds, _p(dst), dst.stride(0), rows, cols), d)
    elif src.dtype == torch.float32 and dst.dtype == BF16:
        _lib.check(_lib.load().crab_cast_rows_f32_bf16(_lib.ctx(d), _stream(), _p(src), lds, _p(dst), dst.stride(0), rows, cols), d)
    else:
        raise _lib.CrabHipError(f"cast_rows: {src.dtype} -> {dst.dtype} is not supported")
    return dst


def rope_table(max_pos: int, head_dim: int, theta: float, device) -> torch.Tensor:
    tab = torch.empty((max_pos, head_dim // 2, 2), device=device, dtype=torch.float32)
    d = _dev(tab)
    _lib.check(_lib.load().crab_rope_table(_lib.ctx(d), _stream(), _p(tab), max_pos, head_dim, theta), d)
    return tab


def qkv_rope_split(qkv: torch.Tensor, rope_tab: Optional[torch.Tensor], k_cache: Optional[torch.Tensor],
                   v_cache: Optional[torch.Tensor], vt: Optional[torch.Tensor], B: int, S: int, H: int, Hk: int, d_: int,
                   Tmax: int, pos0: int = 0, pos_dev: Optional[torch.Tensor] = None, pos_ids: Optional[torch.Tensor] = None,
                   row_off: Optional[torch.Tensor] = None):
    """pos_ids (int32 [B, S]): explicit rotary positions (forward()'s position_ids); the cache slot stays pos0 + s.
    row_off (int32 [B]): the ragged form - sequence b is rotated at slot - row_off[b] (crab_qkv_rope_split_ragged)."""
    d = _dev(qkv)
    vt_ld = vt.stride(-2) if vt is not None else 0
    if row_off is not None:
        if pos_ids is not None:
            raise ValueError("qkv_rope_split: pos_ids and row_off are alternatives")
        _lib.check(_lib.load().crab_qkv_rope_split_ragged(_lib.ctx(d), _stream(), _p(qkv), qkv.stride(0), _p(rope_tab), _p(k_cache), _p(v_cache),
                                                          _p(vt), vt_ld, B, S, H, Hk, d_, Tmax, pos0, _p(pos_dev), _p(row_off)), d)
        return
    if pos_ids is not None and (pos_ids.dtype != torch.int32 or pos_ids.dim() != 2 or pos_ids.stride(1) != 1):
        raise ValueError("pos_ids must be an int32 [B, S] tensor with contiguous rows")
    _lib.check(_lib.load().crab_qkv_rope_split_ids(_lib.ctx(d), _stream(), _p(qkv), qkv.stride(0), _p(rope_tab), _p(k_cache), _p(v_cache),
                                                   _p(vt), vt_ld, B, S, H, Hk, d_, Tmax, pos0, _p(pos_dev), _p(pos_ids),
                                                   pos_ids.stride(0) if pos_ids is not None else 0), d)


def attn_fwd(q, k, vt, o, *, q_strides, k_strides, vt_strides, o_strides, B, H, Hk, Sq, Skv, head_dim, scale, causal=False,
             bias=None, gate=None, kv_start=None, key_mask=None):
    """Strided flash attention; *_strides = (batch, head, row) element strides, o_strides = (batch, row).
    kv_start (int32 [B]): keys below kv_start[b] are masked (left-pad attention_mask); key_mask (int32 [B, >= ceil(Skv / 32)], one
    visibility bit per key: pack_key_mask): any 2-D attention_mask."""
    d = _dev(q)
    a = AttnDesc()
    a.q, a.k, a.vt, a.o = q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr()
    a.q_bs, a.q_hs, a.q_ss = q_strides
    a.k_bs, a.k_hs, a.k_ss = k_strides
    a.vt_bs, a.vt_hs, a.vt_ds = vt_strides
    a.o_bs, a.o_ss = o_strides
    a.bias = bias.data_ptr() if bias is not None else None
    a.gate = gate.data_ptr() if gate is not None else None
    a.kv_start = kv_start.data_ptr() if kv_start is not None else None
    if key_mask is not None:
        if key_mask.dtype != torch.int32 or key_mask.dim() != 2 or key_mask.stride(1) != 1 or key_mask.shape[0] != B:
            raise ValueError("key_mask must be an int32 [B, words] tensor with contiguous rows (ops.pack_key_mask)")
        a.key_mask, a.key_mask_ld = key_mask.data_ptr(), key_mask.stride(0)
    a.B, a.H, a.Hk, a.Sq, a.Skv, a.head_dim, a.causal, a.scale = B, H, Hk, Sq, Skv, head_dim, 1 if causal else 0, scale
    _lib.check(_lib.load().crab_attn_fwd(_lib.ctx(d), _stream(), C.byref(a)), d)
    return o


def pack_key_mask(mask: torch.Tensor) -> torch.Tensor:
    """2-D attention_mask [B, T] (non-zero = attend) -> int32 [B, ceil(T / 32)] visibility words, bit (j & 31) of word j >> 5 = key j
    (crab_attn_desc.key_mask's layout), on the mask's device."""
    m = mask.reshape(mask.shape[0], -1).ne(0)
    B, T = m.shape
    W = (T + 31) // 32
    mp = torch.zeros(B, W * 32, dtype=torch.int64, device=m.device)
    mp[:, :T] = m
    v = (mp.view(B, W, 32) << torch.arange(32, device=m.device)).sum(-1)
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32).contiguous()


def attn_decode(q, k_cache, v_cache, o, B, H, Hk, head_dim, Tmax, ctx_len, scale, ctx_dev=None, kv_start=None, key_mask=None):
    d = _dev(q)
    if key_mask is not None:                          # general attention_mask over the cache rows: its own (rare-path) kernel
        if kv_start is not None:
            raise ValueError("attn_decode: key_mask already carries the left padding, pass one of the two")
        if key_mask.dtype != torch.int32 or key_mask.dim() != 2 or key_mask.stride(1) != 1 or key_mask.shape[0] != B:
            raise ValueError("key_mask must be an int32 [B, words] tensor with contiguous rows (ops.pack_key_mask)")
        _lib.check(_lib.load().crab_attn_decode_keymask(_lib.ctx(d), _stream(), _p(q), q.stride(0), _p(k_cache), _p(v_cache), _p(o), o.stride(0),
                                                        B, H, Hk, head_dim, Tmax, ctx_len, _p(ctx_dev), scale, _p(key_mask), key_mask.stride(0)), d)
        return o
    prof = PROFILER
    sample = prof is not None and prof.decode_eager and prof.decode_ctx > 0 and not torch.cuda.is_current_stream_capturing()
    if sample:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().crab_attn_decode_masked(_lib.ctx(d), _stream(), _p(q), q.stride(0), _p(k_cache), _p(v_cache), _p(o), o.stride(0),
                                                   B, H, Hk, head_dim, Tmax, ctx_len, _p(ctx_dev), scale, _p(kv_start)), d)
    if sample:
        e1.record()
        # algorithmic bytes (DESIGN.md kernel table): every live K and V row once + q in + o out
        nbytes = 2.0 * B * prof.decode_ctx * Hk * head_dim * 2 + 2.0 * B * H * head_dim * 2
        G = H // Hk                                   # mirror of the kernel choice in csrc/attn.hip (crab_attn_decode)
        gqa = head_dim == 128 and G in (2, 4, 7, 8) and B * Hk >= 256
        prof.records.append((f"attn_decode_gqa_kernel<{head_dim},{G}>" if gqa else f"attn_decode_kernel<{head_dim}>", nbytes, e0, e1))
    return o


ATTN_SPLIT_BELOW = _lib.header_const("CRAB_ATTN_SPLIT_BELOW")


def attn_decode_rope_bytes(B: int, H: int, d: int) -> int:
    return int(_lib.load().crab_attn_decode_rope_workspace(B, H, d))


def attn_decode_rope_workspace(B: int, H: int, d: int, device) -> torch.Tensor:
    """Zero-filled workspace of crab_attn_decode_rope (the tickets at its end must start at zero; the kernel leaves them zero)."""
    return torch.zeros((int(_lib.load().crab_attn_decode_rope_workspace(B, H, d)),), device=device, dtype=torch.uint8)


def attn_decode_rope(qkv, rope_tab, k_cache, v_cache, o, B, H, Hk, head_dim, Tmax, pos0, scale, pos_dev=None, workspace=None):
    """RoPE of q / new k + KV append + decode attention over keys 0 .. pos from the raw packed q|k|v rows (small batch)."""
    d = _dev(qkv)
    _lib.check(_lib.load().crab_attn_decode_rope(_lib.ctx(d), _stream(), _p(qkv), qkv.stride(0), _p(rope_tab), _p(k_cache), _p(v_cache), _p(o),
                                                 o.stride(0), B, H, Hk, head_dim, Tmax, pos0, _p(pos_dev), scale, _p(workspace),
                                                 workspace.numel() if workspace is not None else 0), d)
    return o


def enc_layer(kind: str, w, io, device):
    """crab_clip_layer / crab_beats_layer / crab_qformer_layer: one encoder layer's launch sequence in one call."""
    d = device.index or 0
    fn = getattr(_lib.load(), f"crab_{kind}_layer")
    _lib.check(fn(_lib.ctx(d), _stream(), C.byref(w), C.byref(io)), d)


def llama_layers(table, n_layers: int, io, device):
    """crab_llama_layers: every layer of a decoder stack (prefill when io.vt is set, one decode step otherwise) in one call."""
    d = device.index or 0
    _lib.check(_lib.load().crab_llama_layers(_lib.ctx(d), _stream(), table, n_layers, C.byref(io)), d)


def swiglu(gu: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = _dev(gu)
    M, I2 = gu.shape
    if out is None:
        out = torch.empty((M, I2 // 2), device=gu.device, dtype=BF16)
    _lib.check(_lib.load().crab_swiglu(_lib.ctx(d), _stream(), _p(gu), gu.stride(0), _p(out), out.stride(0), M, I2 // 2), d)
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None, suppress: int = -1) -> torch.Tensor:
    d = _dev(logits)
    assert logits.dtype == torch.float32
    B, V = logits.shape
    if out is None:
        out = torch.empty((B,), device=logits.device, dtype=torch.int64)
    _lib.check(_lib.load().crab_argmax(_lib.ctx(d), _stream(), _p(logits), logits.stride(0), _p(out), B, V, suppress), d)
    return out


def im2col_patch(x: torch.Tensor, P: int, ldo: int) -> torch.Tensor:
    """x [N,C,H,W] fp32/bf16 contiguous -> [N*gh*gw, ldo] bf16 patches (k = c,ky,kx)."""
    d = _dev(x)
    x = x.contiguous()
    N, Cc, Hh, Ww = x.shape
    out = torch.empty((N * (Hh // P) * (Ww // P), ldo), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_im2col_patch(_lib.ctx(d), _stream(), _p(x), 1 if x.dtype == torch.float32 else 0, _p(out), ldo, N, Cc, Hh, Ww, P), d)
    return out


def clip_embed_ln(patch, cls, pos, lnw, lnb, N, P, D, eps):
    d = _dev(patch)
    out = torch.empty((N * (P + 1), D), device=patch.device, dtype=BF16)
    wf = lnw.dtype == torch.float32
    if wf != (lnb.dtype == torch.float32):
        raise _lib.CrabHipError("clip_embed_ln: pre_layrnorm weight and bias must have the same storage")
    _lib.check(_lib.load().crab_clip_embed_ln_p(_lib.ctx(d), _stream(), _p(patch), _p(cls), _p(pos), _p(lnw), _p(lnb), 1 if wf else 0, _p(out),
                                                N, P, D, eps), d)
    return out


def beats_posconv_pad(x: torch.Tensor, B: int, n: int, E: int, G: int, Kc: int) -> torch.Tensor:
    d = _dev(x)
    out = torch.empty((G, B, n + Kc - 1, E // G), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_beats_posconv_pad(_lib.ctx(d), _stream(), _p(x), _p(out), B, n, E, G, Kc), d)
    return out


def beats_relpos_bias(table: torch.Tensor, n: int, H: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    d = _dev(table)
    out = torch.empty((H, n, n), device=table.device, dtype=torch.float32)
    _lib.check(_lib.load().crab_beats_relpos_bias(_lib.ctx(d), _stream(), _p(table), _p(out), n, H, num_buckets, max_distance), d)
    return out


def beats_gru_gate(q: torch.Tensor, gw, gb, grep_a, B: int, n: int, H: int, d_: int) -> torch.Tensor:
    d = _dev(q)
    out = torch.empty((B, H, n), device=q.device, dtype=torch.float32)
    _lib.check(_lib.load().crab_beats_gru_gate(_lib.ctx(d), _stream(), _p(q), q.stride(0), _p(gw), _p(gb), _p(grep_a), _p(out), B, n, H, d_), d)
    return out


def copy_rows(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, lds: Optional[int] = None, ldd: Optional[int] = None):
    """dst[r, :cols] = src[r, :cols]; lds/ldd override the row strides (e.g. gather every S-th row)."""
    d = _dev(src)
    _lib.check(_lib.load().crab_copy_rows(_lib.ctx(d), _stream(), _p(src), src.stride(0) if lds is None else lds, _p(dst),
                                          dst.stride(0) if ldd is None else ldd, rows, cols), d)


def copy_rows_batched(src, lds, sbs, dst, ldd, dbs, batch, rows, cols):
    d = _dev(src)
    _lib.check(_lib.load().crab_copy_rows_batched(_lib.ctx(d), _stream(), _p(src), lds, sbs, _p(dst), ldd, dbs, batch, rows, cols), d)


def greedy_select(logits, cur_ids, out_ids, step_dev, finished, eos_id: int, pad_id: int, min_new_tokens: int):
    d = _dev(logits)
    B, V = logits.shape
    _lib.check(_lib.load().crab_greedy_select(_lib.ctx(d), _stream(), _p(logits), logits.stride(0), B, V, _p(cur_ids), _p(out_ids),
                                              out_ids.stride(0), _p(step_dev), _p(finished), eos_id, pad_id, min_new_tokens), d)


def sample_select(logits, cur_ids, out_ids, step_dev, finished, eos_id: int, pad_id: int, min_new_tokens: int, temperature: float, top_k: int,
                  top_p: float, seed: int):
    """crab_sample_select: temperature -> top-k -> top-p -> draw (HF sample mode), device-resident like greedy_select."""
    d = _dev(logits)
    B, V = logits.shape
    _lib.check(_lib.load().crab_sample_select(_lib.ctx(d), _stream(), _p(logits), logits.stride(0), B, V, _p(cur_ids), _p(out_ids),
                                              out_ids.stride(0), _p(step_dev), _p(finished), eos_id, pad_id, min_new_tokens, float(temperature),
                                              int(top_k), float(top_p), int(seed) & 0xFFFFFFFFFFFFFFFF), d)


def advance(pos_dev, step_dev):
    d = _dev(pos_dev)
    _lib.check(_lib.load().crab_advance(_lib.ctx(d), _stream(), _p(pos_dev), _p(step_dev)), d)


def im2col3x3(x: torch.Tensor, B: int, h: int, w: int) -> torch.Tensor:
    """x [B*h*w, C] token-major -> [B*h*w, 9*C] (Conv2d k=3 pad=1 operand)."""
    d = _dev(x)
    Cc = x.shape[1]
    out = torch.empty((B * h * w, 9 * Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_im2col3x3(_lib.ctx(d), _stream(), _p(x), _p(out), B, h, w, Cc), d)
    return out


def pixel_shuffle2x(g: torch.Tensor, bias: Optional[torch.Tensor], h: int, w: int, Co: int) -> torch.Tensor:
    d = _dev(g)
    out = torch.empty((4 * h * w, Co), device=g.device, dtype=BF16)
    _lib.check(_lib.load().crab_pixel_shuffle2x(_lib.ctx(d), _stream(), _p(g), _p(bias), _p(out), h, w, Co), d)
    return out


def bilinear(x: torch.Tensor, strides, Cc: int, h: int, w: int, out: torch.Tensor, alpha: float = 1.0, beta: float = 0.0):
    """x element (c,y,x) at x.data + c*strides[0] + y*strides[1] + x*strides[2]; out fp32 [C,H,W] (accumulates when beta != 0)."""
    d = _dev(x)
    _lib.check(_lib.load().crab_bilinear(_lib.ctx(d), _stream(), _p(x), 1 if x.dtype == torch.float32 else 0, strides[0], strides[1], strides[2],
                                         Cc, h, w, _p(out), out.shape[-2], out.shape[-1], alpha, beta), d)
    return out


def dense_pe(G: torch.Tensor, h: int, w: int) -> torch.Tensor:
    d = _dev(G)
    assert G.dtype == torch.float32 and G.is_contiguous()
    Fq = G.shape[1]
    pe = torch.empty((h * w, 2 * Fq), device=G.device, dtype=BF16)
    _lib.check(_lib.load().crab_dense_pe(_lib.ctx(d), _stream(), _p(G), _p(pe), h, w, Fq), d)
    return pe


def add_rows(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m] = a[m] + b[m % b.shape[0]] (2-D, bf16)."""
    d = _dev(a)
    M, D = a.shape
    if out is None:
        out = torch.empty((M, D), device=a.device, dtype=BF16)
    _lib.check(_lib.load().crab_add_rows(_lib.ctx(d), _stream(), _p(a), a.stride(0), _p(b), b.stride(0), b.shape[0], _p(out), out.stride(0), M, D), d)
    return out


def mask_gate(prev: torch.Tensor, src: torch.Tensor):
    d = _dev(src)
    _lib.check(_lib.load().crab_mask_gate(_lib.ctx(d), _stream(), _p(prev), prev.stride(0), prev.shape[1], _p(src), src.stride(0), src.shape[0], src.shape[1]), d)
    return src


def group_mean(x: torch.Tensor, G: int, T: int, scale: float) -> torch.Tensor:
    """out[g] = scale * sum_{k<T} x[g*T + k]  (bf16 rows)."""
    d = _dev(x)
    D = x.shape[1]
    out = torch.empty((G, D), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_group_mean(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(out), out.stride(0), G, T, D, scale), d)
    return out


def act_inplace(x: torch.Tensor, act: str):
    d = _dev(x)
    assert x.is_contiguous()
    _lib.check(_lib.load().crab_act_inplace(_lib.ctx(d), _stream(), _p(x), x.numel(), ACT[act]), d)
    return x


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 on device (harness-side cast of modality inputs)."""
    if x.dtype == BF16:
        return x
    d = _dev(x)
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_cast_f32_bf16(_lib.ctx(d), _stream(), _p(x), _p(out), x.numel()), d)
    return out


def sync():
    d = torch.cuda.current_device()
    _lib.check(_lib.load().crab_sync(_lib.ctx(d), _stream()), d)


# ------------------------------------------------------------------------------------------------ VQGAN (csrc/vq_ops.hip)
def im2col3x3_strided(x: torch.Tensor, B: int, h: int, w: int, stride: int, pad_top: int, pad_left: int, oh: int, ow: int) -> torch.Tensor:
    """x [B*h*w, C] token-major -> [B*oh*ow, 9*C] window operand (zero outside the image)."""
    d = _dev(x)
    Cc = x.shape[1]
    out = torch.empty((B * oh * ow, 9 * Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_im2col3x3_strided(_lib.ctx(d), _stream(), _p(x), _p(out), B, h, w, Cc, stride, pad_top, pad_left, oh, ow), d)
    return out


def groupnorm(x: torch.Tensor, B: int, HW: int, G: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6, swish: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B*HW, C] token-major bf16 -> GroupNorm(G, C) (+ swish)."""
    d = _dev(x)
    Cc = x.shape[1]
    if out is None:
        out = torch.empty_like(x)
    need = int(_lib.load().crab_groupnorm_workspace(B, HW, G))
    ws = torch.empty((need,), device=x.device, dtype=torch.uint8)
    wf = weight.dtype == torch.float32
    if (bias.dtype == torch.float32) != wf or (not wf and weight.dtype != BF16):
        raise _lib.CrabHipError("groupnorm: weight and bias are both bf16 or both fp32")
    _lib.check(_lib.load().crab_groupnorm_p(_lib.ctx(d), _stream(), _p(x), _p(out), B, HW, Cc, G, eps, _p(weight), _p(bias), 1 if wf else 0,
                                            1 if swish else 0, _p(ws), need), d)
    return out


def upsample_nearest2x(x: torch.Tensor, B: int, h: int, w: int) -> torch.Tensor:
    d = _dev(x)
    Cc = x.shape[1]
    out = torch.empty((B * 4 * h * w, Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_upsample_nearest2x(_lib.ctx(d), _stream(), _p(x), _p(out), B, h, w, Cc), d)
    return out


def softmax_rows(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x fp32 [M,N] -> bf16 softmax(scale * x) per row (out: a [M, N] view of a wider buffer, unit column stride)."""
    d = _dev(x)
    assert x.dtype == torch.float32 and x.stride(1) == 1
    M, N = x.shape
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16)
    assert out.dtype == BF16 and tuple(out.shape) == (M, N) and out.stride(1) == 1
    _lib.check(_lib.load().crab_softmax_rows(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(out), out.stride(0), M, N, scale), d)
    return out


def row_sqnorm(e: torch.Tensor) -> torch.Tensor:
    d = _dev(e)
    N, D = e.shape
    out = torch.empty((N,), device=e.device, dtype=torch.float32)
    _lib.check(_lib.load().crab_row_sqnorm(_lib.ctx(d), _stream(), _p(e), e.stride(0), N, D, _p(out)), d)
    return out


def split3(x: torch.Tensor, pattern: int) -> torch.Tensor:
    """fp32 [M, C] -> bf16 [M, 3 * round_up(C, 8)] split operand: x = hi + lo; pattern 0 = [hi | lo | hi] (A side), 1 = [hi | hi | lo] (B side).
    gemm(split3(x, 0), split3(w, 1)) = x_hi.w_hi + x_lo.w_hi + x_hi.w_lo with fp32 accumulation (crab_split3)."""
    d = _dev(x)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise _lib.CrabHipError("split3: fp32 [M, C] rows expected")
    M, Cc = x.shape
    Cp = (Cc + 7) // 8 * 8
    out = torch.empty((M, 3 * Cp), device=x.device, dtype=BF16)
    _lib.check(_lib.load().crab_split3(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(out), out.stride(0), M, Cc, pattern), d)
    return out


def groupnorm_f32(x: torch.Tensor, B: int, HW: int, G: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6, swish: bool = False) -> torch.Tensor:
    """GroupNorm (+ swish) with fp32 input, output and parameters (the precise VQGAN encoder)."""
    d = _dev(x)
    assert x.dtype == weight.dtype == bias.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    need = int(_lib.load().crab_groupnorm_workspace(B, HW, G))
    ws = torch.empty((need,), device=x.device, dtype=torch.uint8)
    _lib.check(_lib.load().crab_groupnorm_f32(_lib.ctx(d), _stream(), _p(x), _p(out), B, HW, x.shape[1], G, eps, _p(weight), _p(bias), 1 if swish else 0, _p(ws), need), d)
    return out


def add_bias_f32(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    d = _dev(x)
    assert x.dtype == bias.dtype == torch.float32 and x.stride(1) == 1 and bias.numel() == x.shape[1]
    _lib.check(_lib.load().crab_add_bias_f32(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(bias), x.shape[0], x.shape[1]), d)
    return x


def softmax_rows_f32(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    d = _dev(x)
    assert x.dtype == torch.float32 and x.stride(1) == 1
    M, N = x.shape
    out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().crab_softmax_rows_f32(_lib.ctx(d), _stream(), _p(x), x.stride(0), _p(out), out.stride(0), M, N, scale), d)
    return out


def row_sqnorm_f32(e: torch.Tensor) -> torch.Tensor:
    d = _dev(e)
    assert e.dtype == torch.float32 and e.stride(1) == 1
    N, D = e.shape
    out = torch.empty((N,), device=e.device, dtype=torch.float32)
    _lib.check(_lib.load().crab_row_sqnorm_f32(_lib.ctx(d), _stream(), _p(e), e.stride(0), N, D, _p(out)), d)
    return out


def vq_nearest_f32(z: torch.Tensor, e: torch.Tensor, e2: torch.Tensor, offset: int = 0) -> torch.Tensor:
    """z fp32 [M, D] latents, e fp32 [N, D] codebook, e2 = row_sqnorm_f32(e) -> int64 [M]: offset + first argmin_n (|z|^2 + e2[n]) - 2 z . e_n,
    every term in fp32 (quantize.py:286-290): codebook ids are index work."""
    d = _dev(z)
    if z.dtype != torch.float32 or e.dtype != torch.float32 or e2.dtype != torch.float32 or z.stride(1) != 1 or e.stride(1) != 1:
        raise _lib.CrabHipError("vq_nearest_f32: fp32 row-major operands")
    M, D = z.shape
    N = e.shape[0]
    out = torch.empty((M,), device=z.device, dtype=torch.int64)
    need = int(_lib.load().crab_vq_nearest_f32_workspace(M, N))
    ws = torch.empty((need,), device=z.device, dtype=torch.uint8)
    _lib.check(_lib.load().crab_vq_nearest_f32(_lib.ctx(d), _stream(), _p(z), z.stride(0), _p(e), e.stride(0), _p(e2), M, N, D, _p(out), offset, _p(ws), need), d)
    return out


def vq_argmin(dots: torch.Tensor, e2: torch.Tensor, offset: int = 0) -> torch.Tensor:
    """dots fp32 [M,N] = z . e^T, e2 fp32 [N] -> int64 [M]: offset + first argmin_n (e2[n] - 2 dots[m,n])."""
    d = _dev(dots)
    assert dots.dtype == torch.float32 and e2.dtype == torch.float32 and dots.stride(1) == 1
    M, N = dots.shape
    out = torch.empty((M,), device=dots.device, dtype=torch.int64)
    _lib.check(_lib.load().crab_vq_argmin(_lib.ctx(d), _stream(), _p(dots), dots.stride(0), _p(e2), M, N, _p(out), offset), d)
    return out


def mask_labels(pred: torch.Tensor) -> torch.Tensor:
    """pred [C, H, W] fp32 (a SegModule mask) -> uint8 [H, W]: 0 / 255 by sigmoid > 0.5 for C == 1, the argmax class index otherwise
    (what the reference's eval loops write to the PNG)."""
    d = _dev(pred)
    if pred.dtype != torch.float32 or pred.dim() != 3:
        raise _lib.CrabHipError("mask_labels: fp32 [C, H, W] expected")
    pred = pred.contiguous()
    C_, H, W = pred.shape
    out = torch.empty((H, W), device=pred.device, dtype=torch.uint8)
    _lib.check(_lib.load().crab_mask_labels(_lib.ctx(d), _stream(), _p(pred), C_, H * W, _p(out)), d)
    return out
