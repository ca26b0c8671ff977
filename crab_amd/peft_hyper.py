"""hyper-LoRA adapter layer: host-side mirror of the reference's forked PEFT (`peft_hyper/`).

Reference interface kept (same names / argument meaning):
  * `LoraConfig(r, lora_alpha, lora_dropout, lora_nums, target_modules, ...)`   peft_hyper/tuners/lora.py:42-83
  * `get_peft_model(model, peft_config)`                                        peft_hyper/mapping.py:182-204
  * `PeftModelForCausalLM.generate(**kwargs)`, `.get_model()`, state-dict keys prefixed `base_model.model.`
    (peft_hyper/peft_model.py:584-593; key names verified against the reference in tests/golden)
  * `Linear` with `lora_route`, `lora_A`, `lora_B{i}` sub-modules and `scaling = lora_alpha / r`
    (peft_hyper/tuners/lora.py:260-369)

MI355X design: the adapter cannot be merged (routing is per token), so it is FUSED instead:
  y = x W^T + b + sum_i softmax(x R^T)_i (x A^T) B_i^T * s      (lora.py:341-350)
    = [x | u] . [W | Bcat]^T + b,   u = s * (softmax(xR^T) (x) xA^T) in R^{nl*r}
i.e. the LoRA update is a K-extension of the base GEMM by nl*r (=24, padded to 32) columns.  All Linear
modules that consume the same activation (q/k/v; gate/up) are packed into ONE weight matrix, ONE skinny
[R;A] matrix and ONE block-structured Bcat matrix (`PackedLinearGroup`), so a decoder layer is
4 skinny GEMMs + 4 routing-mix launches + 4 two-segment MFMA GEMMs instead of the reference's 7 x ~13
eager ops.  The nn.Parameters exposed under the reference's names are VIEWS into the packed buffers:
`load_state_dict` of a reference checkpoint fills the packed operands in place, no repacking pass.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops

BF16 = torch.bfloat16


def _pad(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class LoraConfig:
    """peft_hyper/tuners/lora.py:42-83 (fields used on the inference path)."""
    r: int = 8
    lora_alpha: int = 16
    lora_dropout: float = 0.05          # identity in eval
    lora_nums: int = 3
    target_modules: Sequence[str] = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "down_proj", "up_proj")
    task_type: str = "CAUSAL_LM"
    inference_mode: bool = False
    modules_to_save: Optional[List[str]] = None


class _W(nn.Module):
    """Bias-free sub-linear holding only `.weight` (lora_route / lora_A / lora_B{i}, lora.py:286-290)."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)


class Linear(nn.Module):
    """One projection of a PackedLinearGroup.  Exposes `weight` (+`bias`) and, once adapted, `lora_route`,
    `lora_A`, `lora_B0..` exactly like peft_hyper/tuners/lora.py:260-292.  Holds no storage of its own."""

    def __init__(self, group: "PackedLinearGroup", index: int):
        super().__init__()
        self._group = [group]            # list: keep the group out of the module tree
        self._index = index
        n0, n1 = group.row_range(index)
        self.in_features, self.out_features = group.K, n1 - n0
        self.weight = nn.Parameter(group.rows(group.W, index), requires_grad=False)
        if group.bias is not None:
            self.bias = nn.Parameter(group.rows(group.bias, index), requires_grad=False)
        else:
            self.register_parameter("bias", None)
        self.r = 0

    def _attach_lora(self):
        g = self._group[0]
        p, r, nl = self._index, g.r, g.nl
        n0, n1 = g.row_range(p)
        self.r, self.lora_num, self.scaling = r, nl, g.scaling
        base = p * (nl + r)
        self.lora_route = _W(g.RA[base:base + nl])
        self.lora_A = _W(g.RA[base + nl:base + nl + r])
        for i in range(nl):
            setattr(self, f"lora_B{i}", _W(g.rows(g.B2, p)[:, p * nl * r + i * r: p * nl * r + (i + 1) * r]))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Standalone use (tests / odd callers): runs the group's fused path and slices this projection."""
        g = self._group[0]
        n0, n1 = g.row_range(self._index)
        shp = x.shape
        y = g(x.reshape(-1, shp[-1]))
        return g.rows(y.t(), self._index).t().reshape(*shp[:-1], n1 - n0)


class PackedLinearGroup:
    """Linears sharing one input, packed for the fused hyper-LoRA GEMM.  Not an nn.Module: its member
    `Linear`s own the Parameters (as views).  `__call__(x[M,K]) -> y[M, sum N_i]` launches the HIP path."""

    def __init__(self, names: Sequence[str], in_features: int, out_features: Sequence[int], bias: bool, device, dtype=BF16,
                 interleave: bool = False):
        """interleave: the rows of two equal-width members alternate (gate_0, up_0, gate_1, ...) instead of being
        stacked, so that the GEMM epilogue sees (gate_i, up_i) in adjacent columns and can apply SwiGLU itself
        (`__call__(..., act="swiglu_pair")`).  The member Parameters are then stride-2 row views."""
        if dtype != BF16:
            raise ValueError("the MI355X path stores weights in bfloat16")
        if interleave and (len(names) != 2 or out_features[0] != out_features[1]):
            raise ValueError("interleave needs two members of equal width")
        self.interleave = interleave
        self.prof_class = None            # ops.KernelProfiler class tag of this group's launches (decoder.py sets "decoder")
        self.names = list(names)
        self.K = in_features
        self.outs = list(out_features)
        self.N = sum(self.outs)
        self.W = torch.zeros((self.N, in_features), device=device, dtype=dtype)
        self.bias = torch.zeros((self.N,), device=device, dtype=dtype) if bias else None
        self.r = self.nl = 0
        self.scaling = 1.0
        self.RA = self.B2 = None
        self.linears = [Linear(self, i) for i in range(len(names))]

    def row_range(self, i: int):
        n0 = sum(self.outs[:i])
        return n0, n0 + self.outs[i]

    def rows(self, t: torch.Tensor, i: int) -> torch.Tensor:
        """The rows of packed tensor `t` ([N, ...]) that belong to member i (a view)."""
        if self.interleave:
            return t[i::2]
        n0, n1 = self.row_range(i)
        return t[n0:n1]

    def attach_lora(self, r: int, lora_alpha: int, lora_nums: int):
        nproj = len(self.names)
        self.r, self.nl, self.scaling = r, lora_nums, lora_alpha / r
        dev = self.W.device
        self.t_cols = _pad(nproj * (lora_nums + r), 16)
        self.u_cols = _pad(nproj * lora_nums * r, 32)
        self.RA = torch.zeros((self.t_cols, self.K), device=dev, dtype=BF16)
        self.B2 = torch.zeros((self.N, self.u_cols), device=dev, dtype=BF16)
        for lin in self.linears:
            lin._attach_lora()

    def rebind(self, fn=None):
        """Re-point the member Parameters at the packed buffers, after applying `fn` (the tensor map of nn.Module._apply:
        a device move) to the buffers.  nn.Module._apply replaces every view Parameter with an independent tensor, which
        would leave W / RA / B2 behind; UnifiedForCausalLM._apply calls this for every group afterwards.  A dtype change is
        refused: the HIP path computes on bf16 storage only."""
        def mv(t):
            if t is None or fn is None:
                return t
            r = fn(t)
            if r.dtype != BF16:
                raise TypeError("crab_amd keeps decoder weights in bfloat16: .float() / .half() / .to(dtype) are not supported")
            return r
        self.W, self.bias, self.RA, self.B2 = mv(self.W), mv(self.bias), mv(self.RA), mv(self.B2)
        for i, lin in enumerate(self.linears):
            lin.weight = nn.Parameter(self.rows(self.W, i), requires_grad=False)
            if self.bias is not None:
                lin.bias = nn.Parameter(self.rows(self.bias, i), requires_grad=False)
            if self.RA is not None:
                lin._attach_lora()

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 t_buf: Optional[torch.Tensor] = None, u_buf: Optional[torch.Tensor] = None, post_norm=None, act: str = "none", rope=None, route_next=None, u_ready=None, info=None,
                 rope_row_off=None) -> torch.Tensor:
        """y = group(x) (+residual).  post_norm = (rms_weight, eps, h_out): additionally h_out = rmsnorm(y) * rms_weight
        (the LlamaRMSNorm that follows o_proj / down_proj), fused into the GEMM epilogue in the decode regime.
        act = "swiglu_pair" (interleaved groups): returns silu(member0(x)) * member1(x), [M, N/2]."""
        M = x.shape[0]
        if act == "swiglu_pair" and not self.interleave:
            raise ValueError("swiglu_pair needs an interleaved group")
        # route_next = (group, u_out): this GEMM's fused post-norm also evaluates the router of the NEXT group on the
        # normalised rows; that group is then called with u_ready=u_out and skips its own router launches
        route = None
        if route_next is not None and route_next[0].RA is not None and post_norm is not None and M <= ops.DECODE_MAX_ROWS:
            ng, nu = route_next
            route = (ng.RA, len(ng.names), ng.nl, ng.r, ng.u_cols, ng.scaling, nu[:M, :ng.u_cols])
        if self.RA is None:
            return ops.gemm(x, self.W, bias=self.bias, residual=residual, out=out, post_norm=post_norm, act=act, rope=rope, route=route, info=info,
                            prof_class=self.prof_class, rope_row_off=rope_row_off)
        if u_ready is None and M <= 16 and post_norm is not None and len(self.names) == 1 and ops.ROWFIN and ops.rowfin_lora_ok(self.nl, self.r, self.N):
            # the reference's batch sizes (M <= 16), o_proj / down_proj: no router launches - the [R;A] rows ride on the projection's
            # launch and the update is applied by the wide layer tail (csrc/rowfin.hip) together with the residual row, its RMSNorm and
            # the next group's router
            return ops.gemm(x, self.W, bias=self.bias, residual=residual, out=out, post_norm=post_norm, act=act, route=route,
                            lora_self=(self.RA, self.nl, self.r, self.scaling, self.B2), prof_class=self.prof_class)
        if u_ready is not None:
            u = u_ready[:M, :self.u_cols]
        else:
            u = u_buf[:M, :self.u_cols] if u_buf is not None else torch.empty((M, self.u_cols), device=x.device, dtype=BF16)
            # route logits | lora_A(x) -> softmax mix, K split over blocks (skinny.hip); t_buf is the partial-sum workspace
            ops.hyperlora_route(x, self.RA, len(self.names), self.nl, self.r, self.u_cols, self.scaling, out=u, workspace=t_buf)
        return ops.gemm(x, self.W, bias=self.bias, residual=residual, x2=u, w2=self.B2, out=out, post_norm=post_norm, act=act, rope=rope,
                        route=route, info=info, prof_class=self.prof_class, rope_row_off=rope_row_off)

    def routes_ahead(self, M: int) -> bool:
        """True when a producer GEMM may evaluate this group's router in its fused post-norm epilogue (decode regime)."""
        return self.RA is not None and M <= ops.DECODE_MAX_ROWS


class PeftModelForCausalLM(nn.Module):
    """peft_hyper/peft_model.py: wraps the model as `base_model.model`, so checkpoint keys read
    `base_model.model.model.layers.N.self_attn.q_proj.lora_A.weight`."""

    def __init__(self, model: nn.Module, peft_config: LoraConfig):
        super().__init__()
        self.peft_config = peft_config
        holder = nn.Module()
        holder.model = model
        self.base_model = holder

    def get_model(self):
        return self.base_model.model.get_model()

    def generate(self, **kwargs):
        return self.base_model.model.generate(**kwargs)

    def generate_avs(self, **kwargs):
        return self.base_model.model.generate_avs(**kwargs)

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model.model, name)


def get_peft_model(model: nn.Module, peft_config: LoraConfig) -> PeftModelForCausalLM:
    """peft_hyper/mapping.py:182-204 + LoraModel._find_and_replace (lora.py:118-159): every target Linear gains
    the routed adapter.  Here: each PackedLinearGroup whose members are targets gets its [R;A] / Bcat operands."""
    targets = set(peft_config.target_modules)
    for grp in model.packed_groups():
        hit = [n in targets for n in grp.names]
        if any(hit):
            if not all(hit):
                raise NotImplementedError(f"partial adaptation of a packed group {grp.names} is not supported")
            grp.attach_lora(peft_config.r, peft_config.lora_alpha, peft_config.lora_nums)
    model._invalidate_graphs()
    return PeftModelForCausalLM(model, peft_config)
