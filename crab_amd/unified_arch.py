"""Mirror of reference `models/unified_arch.py`: the mixins that build the encoders/projectors, encode video / audio,
splice modality features into the token-embedding sequence and left-pad the batch.

Same public names and argument meaning:
  UnifiedMetaModel.init_multimodal_modules(...)          unified_arch.py:31-110
  UnifiedMetaModel.encode_video / encode_audio            :113-155
  UnifiedMetaForCausalLM.encode_video / encode_audio / encode_ids      :185-214
  UnifiedMetaForCausalLM.prepare_multimodal_inputs(...)   :217-406
  UnifiedMetaForCausalLM.initialize_MM_tokenizer(...)     :409-459
All tensor arithmetic (encoders, embedding gathers) runs in the HIP library; the splice / left-pad bookkeeping is
row copies into one preallocated [bs, S, D] buffer (the reference builds it with many small torch.cat calls).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .multimodal_encoder import ALProjector, AudioEncoder, VLProjector, VisualEncoder

BF16 = torch.bfloat16
AVS_TASKS = ('ms3', 's4', 'avss', 'ref-avs')


ENC_CHUNK = int(os.environ.get("CRAB_ENC_CHUNK", "96"))      # most clips (modality blocks) per encoder call in prepare_multimodal_inputs


def plan_enc_chunks(n: int, rows_per_block: int, cmax: int) -> List[int]:
    """Sizes of the encoder calls for n equal modality blocks of rows_per_block token rows each, at most cmax blocks per call.  A tower GEMM runs
    on 256 x 256 tiles, one per CU and round of 256 (csrc/gemm_glds.hip): 64 clips x 8 frames x 257 rows are 514 row tiles, and the width-1024
    projections (4 column tiles) then need 8.03 rounds = 9, an 11 % loss on half of the tower's GEMM time (measured: encoder class 0.378 ->
    0.353 of the MFMA peak when the chunking arrived with 64).  So the chunk size is chosen per call: the c in [cmax / 2, cmax] whose whole
    partition (n // c calls of c blocks + the rest) costs the fewest tile rounds over the tower's four projection shapes, weighted by their K."""
    if n <= 0:
        return []
    if rows_per_block <= 0 or n <= max(1, cmax // 2) or os.environ.get("CRAB_ENC_PLAN") == "0":       # "0": fixed-size chunks (A/B runs)
        return [n] if n <= cmax else [cmax] * (n // cmax) + ([n % cmax] if n % cmax else [])

    def rounds(m):                                   # (column tiles, K in units of 1024) of q|k|v, out, fc1, fc2 at width 1024 / 4096
        rt = -(-m * rows_per_block // 256)
        return sum(-(-rt * ct // 256) * ku for ct, ku in ((12, 1), (4, 1), (16, 1), (4, 4)))
    best = None
    for c in range(min(cmax, n), max(1, cmax // 2) - 1, -1):
        k, r = divmod(n, c)
        cost = k * rounds(c) + (rounds(r) if r else 0)
        if best is None or cost < best[0]:
            best = (cost, c)
    c = best[1]
    return [c] * (n // c) + ([n % c] if n % c else [])

class UnifiedMetaModel:

    def init_multimodal_modules(
        self,
        d_model=4096,
        # visual
        vit_ckpt_path=None,
        select_layer_list=[14, 22, 23],
        select_feature='patch',
        image_size=224,
        patch_size=14,
        visual_query_token_nums=32,
        # audio
        BEATs_ckpt_path=None,
        audio_query_token_nums=32,
        # seg
        image_scale_nums=2,
        token_nums_per_scale=3,
        avs_query_num=300,
        num_classes=1,
        query_generator_num_layers=2,
        prompt_embed_dim=256,
        mask_decoder_transformer_depth=2,
        low_res_mask_size=112,
        dice_loss_weight=0.5,
        bce_loss_weight=2.0,
        vit_image_embedding_dim=1024,
        visual_branch=False,
        audio_branch=False,
        segment_branch=False,
        use_vqgan=False,
        # build-side extensions (no checkpoints offline: architectures can be given explicitly)
        clip_config: Optional[Dict] = None,
        beats_config: Optional[Dict] = None,
        bert_config: Optional[Dict] = None,
        vqgan_config: Optional[Dict] = None,
    ):
        dev = self.embed_tokens.weight.device
        if visual_branch:
            image_token_nums = (image_size // patch_size) * (image_size // patch_size)
            self.visual_encoder = VisualEncoder(model_name_or_path=vit_ckpt_path, select_layer_list=select_layer_list,
                                                select_feature=select_feature, config=clip_config, device=dev)
            enc_w = self.visual_encoder.vision_tower.config["hidden_size"]
            self.vl_projector = VLProjector(hidden_size=enc_w, d_model=d_model, depth=2, image_token_nums=image_token_nums,
                                            num_query_token=visual_query_token_nums, num_hidden_layers=2,
                                            bert_config=bert_config, device=dev)
        if audio_branch:
            self.audio_encoder = AudioEncoder(ckpt_path=BEATs_ckpt_path, cfg=beats_config, device=dev)
            enc_w = self.audio_encoder.audio_encoder.cfg.encoder_embed_dim
            self.al_projector = ALProjector(hidden_size=enc_w, d_model=d_model, depth=2, num_query_token=audio_query_token_nums,
                                            num_hidden_layers=2, bert_config=bert_config, device=dev)
        if segment_branch:
            from .seg_module import SegModule
            self.low_res_mask_size = low_res_mask_size
            self.seg_module = SegModule(
                d_model=d_model, prompt_embed_dim=prompt_embed_dim, image_scale_nums=image_scale_nums,
                token_nums_per_scale=token_nums_per_scale, mask_decoder_transformer_depth=mask_decoder_transformer_depth,
                vit_image_embedding_dim=vit_image_embedding_dim, avs_query_num=avs_query_num, num_classes=num_classes,
                query_generator_num_layers=query_generator_num_layers, image_size=image_size, patch_size=patch_size,
                image_embedding_size=(image_size // patch_size), dice_loss_weight=dice_loss_weight,
                bce_loss_weight=bce_loss_weight, device=dev)
        if use_vqgan:
            # unified_arch.py:109-110; `vqgan_config` (build-side extension) overrides the taming f16/16384 architecture
            from .vqgan import MaskEncoder
            vq = dict(vqgan_config or {})
            self.mask_encoder = MaskEncoder(token_shift=32000 + 20, device=dev, ddconfig=vq.get("ddconfig"), n_embed=vq.get("n_embed", 16384),
                                            embed_dim=vq.get("embed_dim", 256))

    def encode_mask(self, mask):
        """unified_arch.py:158-159."""
        return self.mask_encoder(mask)

    def encode_video(self, visual, all_levels: bool = False):
        """unified_arch.py:144-149.  The reference pushes all three CLIP feature levels through the VLProjector and
        consumes only the last (:290); the two dead passes are skipped unless all_levels=True (entries are None)."""
        vit_feature_list = self.visual_encoder(visual)                    # [(b,t*n,d), ...]
        qformer_feature_list = []
        for i, vit_feature in enumerate(vit_feature_list):
            if all_levels or i == len(vit_feature_list) - 1:
                qformer_feature_list.append(self.vl_projector(vit_feature))
            else:
                qformer_feature_list.append(None)
        return vit_feature_list, qformer_feature_list

    def encode_audio(self, audio):
        return self.al_projector(self.audio_encoder(audio))

    def postprocess_seg(self, pred_embeddings, multi_scale_image_feature_list, gt_mask=None, batch_task_names=[]):
        """unified_arch.py:162-176."""
        return self.seg_module(pred_embeddings=pred_embeddings, multi_scale_image_feature_list=multi_scale_image_feature_list,
                               low_res_mask_size=self.low_res_mask_size, gt_mask=gt_mask, batch_task_names=batch_task_names)


class UnifiedMetaForCausalLM:

    KEYS = ['<image>', '<video>', '<audio>']

    def get_model(self) -> UnifiedMetaModel:
        raise NotImplementedError

    def encode_audio(self, audio, batch_first=True):
        if not batch_first:
            audio = audio.unsqueeze(0)
        f = self.get_model().encode_audio(self._to_model_dtype(audio))
        return f if batch_first else f.squeeze(0)

    def encode_video(self, video, batch_first=True):
        if not batch_first:
            video = video.unsqueeze(0)
        vit, qf = self.get_model().encode_video(self._to_model_dtype(video))
        if not batch_first:
            vit = [v.squeeze(0) for v in vit]
            qf = [q.squeeze(0) if q is not None else None for q in qf]
        return vit, qf

    def encode_mask(self, mask, batch_first=False):
        """unified_arch.py:204-210: mask image [3,H,W] (or [b,3,H,W]) -> VQGAN token ids (already shifted into the LLM vocab)."""
        if not batch_first:
            mask = mask.unsqueeze(0)
        indices = self.get_model().encode_mask(mask)
        return indices if batch_first else indices.squeeze(0)

    def encode_ids(self, ids):
        return self.get_model().embed_tokens(ids)

    def _to_model_dtype(self, x: torch.Tensor) -> torch.Tensor:
        """prepare_sample never casts dtype (utils/util.py:33-47); the bf16 model needs bf16 modality inputs
        (SURVEY.md appendix A.8).  Host->device copy + on-device cast."""
        x = x.to(self.device, non_blocking=True)
        return ops.cast_bf16(x) if x.dtype == torch.float32 else x

    def prepare_multimodal_inputs(
        self,
        batch_input_ids,
        batch_labels,
        batch_X_modals,
        batch_task_names=None,
        return_multi_scale_features=False,
        return_gt_mask=False,
    ):
        """unified_arch.py:217-406.  Returns the same dict: input_ids=None, inputs_embeds [bs,S,D], attention_mask,
        labels, position_ids (+ multi_scale_image_features, mask_token_mask, gt_mask on the AVS path).
        The AVS `<image>` is encoded ONCE (the reference encodes it twice, :244 and :297)."""
        return self.prepare_multimodal_inputs_many(
            [{"batch_input_ids": batch_input_ids, "batch_labels": batch_labels, "batch_X_modals": batch_X_modals,
              "batch_task_names": batch_task_names}],
            return_multi_scale_features=return_multi_scale_features, return_gt_mask=return_gt_mask)[0]

    def prepare_multimodal_inputs_many(self, batches, return_multi_scale_features=False, return_gt_mask=False):
        """prepare_multimodal_inputs for SEVERAL collated batches of the eval loop at once (scripts/finetune/inference_hyper_lora.py:1466-1479
        calls generate() once per batch of 8): every batch keeps its own splice, left padding, attention_mask and position_ids - the dict a
        separate call returns - but the <video> / <audio> blocks of ALL batches go through the encoders together (_encode_blocks: equal shapes
        stacked, chunks sized for whole tile rounds), so 56 batches of 8 clips cost the towers what one batch of 448 costs them instead of 56
        small passes.  Rows never interact inside the encoders, so each batch's features are those of its own call up to the kernels'
        choice of tile schedule for a different M.  Returns one dict per batch."""
        device = self.device
        special = self.SPECIAL_TOKEN_2_IDS
        key_ids = {special[k]: k for k in self.KEYS}
        emb_w = self.get_model().embed_tokens.weight
        D = emb_w.shape[1]

        # ---- pass 1 (host): segment plan per sample; modality blocks of every batch are encoded batched per kind
        vids, auds, msks = [], [], []
        recs = []
        for bt in batches:
            batch_input_ids, batch_labels = bt["batch_input_ids"], bt.get("batch_labels")
            batch_X_modals = bt["batch_X_modals"]
            bs = len(batch_input_ids)
            plans = []
            # one device->host transfer for the ids (and one for the labels) of the whole batch, not one per sample
            n_ids = [int(x.numel()) for x in batch_input_ids]
            flat = torch.cat([x.reshape(-1) for x in batch_input_ids]).tolist()
            ids_ls, o = [], 0
            for n in n_ids:
                ids_ls.append(flat[o:o + n])
                o += n
            lab_ls = None
            if batch_labels is not None:
                flat = torch.cat([x.reshape(-1) for x in batch_labels]).tolist()
                lab_ls, o = [], 0
                for n in n_ids:
                    lab_ls.append(flat[o:o + n])
                    o += n
            for i in range(bs):
                ids_l = ids_ls[i]
                segs, pre = [], 0
                for pos, tok in enumerate(ids_l):
                    if tok in key_ids:
                        segs.append(("text", pre, pos))
                        key = key_ids[tok]
                        if key == '<audio>':
                            segs.append(("audio", len(auds)))
                            auds.append(batch_X_modals[i][key])
                        elif key == '<mask>':                                                # VQGAN mask image -> 256 token ids (:303-307)
                            segs.append(("mask", len(msks)))
                            msks.append(batch_X_modals[i][key])
                        else:
                            segs.append(("video", len(vids)))
                            vids.append(batch_X_modals[i][key])
                        pre = pos + 1
                segs.append(("text", pre, len(ids_l)))
                plans.append(segs)
            recs.append((bt, bs, ids_ls, lab_ls, plans))
        vfeat, vvit = self._encode_blocks(vids, video=True, want_vit=return_multi_scale_features)
        afeat, _ = self._encode_blocks(auds, video=False)
        mids = [self.encode_mask(m, batch_first=False) for m in msks]
        return [self._assemble_inputs(bt, bs, ids_ls, lab_ls, plans, vfeat, vvit, afeat, mids, key_ids, emb_w, D, device,
                                      return_multi_scale_features, return_gt_mask)
                for (bt, bs, ids_ls, lab_ls, plans) in recs]

    def _assemble_inputs(self, bt, bs, ids_ls, lab_ls, plans, vfeat, vvit, afeat, mids, key_ids, emb_w, D, device,
                         return_multi_scale_features, return_gt_mask):
        """Pass 2 of prepare_multimodal_inputs for ONE batch: lengths, left padding, the spliced output buffer, mask / labels / positions."""
        special = self.SPECIAL_TOKEN_2_IDS
        batch_X_modals, batch_task_names = bt["batch_X_modals"], bt.get("batch_task_names")
        img_block = {}                                   # sample -> index of its <image> block (multi-scale features)
        if return_multi_scale_features:
            for i, segs in enumerate(plans):
                ids_l = ids_ls[i]
                k = 0
                for pos, tok in enumerate(ids_l):
                    if tok in key_ids:
                        if key_ids[tok] == '<image>' and i not in img_block:
                            img_block[i] = [sg for sg in segs if sg[0] in ("video", "audio", "mask")][k][1]
                        k += 1

        # ---- pass 2: lengths, left padding, one output buffer
        lens = []
        for segs in plans:
            n = 0
            for sg in segs:
                n += (sg[2] - sg[1]) if sg[0] == "text" else (mids[sg[1]] if sg[0] == "mask" else
                                                              (vfeat[sg[1]] if sg[0] == "video" else afeat[sg[1]])).shape[0]
            lens.append(n)
        S = max(lens)
        out = torch.empty((bs, S, D), device=device, dtype=BF16)
        pad_id = self.get_model().pad_token_id
        attn = torch.zeros((bs, S), dtype=torch.int32)
        labels = torch.full((bs, S), -100, dtype=torch.long)
        # every text / pad row of the batch is looked up by ONE embedding launch: tok[i, s] = token id, -1 = row is
        # written by a modality copy (the kernel leaves rows with a negative id untouched)
        tok = np.full((bs, S), -1, dtype=np.int64)
        for i, segs in enumerate(plans):
            off = S - lens[i]
            tok[i, :off] = pad_id                                                        # :344-348
            cur = off
            for sg in segs:
                if sg[0] == "text":
                    n = sg[2] - sg[1]
                    if n:
                        tok[i, cur:cur + n] = ids_ls[i][sg[1]:sg[2]]
                        if lab_ls is not None:
                            labels[i, cur:cur + n] = torch.tensor(lab_ls[i][sg[1]:sg[2]], dtype=torch.long)
                elif sg[0] == "mask":
                    n = mids[sg[1]].shape[0]
                    ops.embedding(mids[sg[1]], emb_w, out=out[i, cur:cur + n])          # encode_ids(indices), labels = indices
                    labels[i, cur:cur + n] = mids[sg[1]].cpu()
                else:
                    f = vfeat[sg[1]] if sg[0] == "video" else afeat[sg[1]]
                    n = f.shape[0]
                    ops.copy_rows(f, out[i, cur:cur + n], n, D)
                cur += n
            attn[i, off:] = 1
        ops.embedding(torch.from_numpy(tok).reshape(-1), emb_w, out=out.view(bs * S, D))
        position_ids = torch.cumsum(attn, dim=-1) - 1
        position_ids[position_ids == -1] = 0                                             # :372-373
        dict_data = {
            'input_ids': None,
            'inputs_embeds': out,
            'attention_mask': attn.to(device),
            'labels': labels.to(device),
            'position_ids': position_ids.to(device),
        }
        if return_multi_scale_features:
            scale = 2
            ms = [[] for _ in range(scale)]
            mtm = torch.zeros((bs, S), dtype=torch.bool)
            mask_ids = {special[m] for m in self.MASK}
            for i in range(bs):
                is_avs = batch_task_names[i] in AVS_TASKS
                ids_l = ids_ls[i]
                if is_avs and i in img_block:
                    for sc in range(scale):
                        ms[sc].append(vvit[img_block[i]][sc])
                else:                                                                      # :235-239 zeros for non-AVS rows
                    for sc in range(scale):
                        ms[sc].append(torch.zeros((256, 1024), device=device, dtype=BF16))
                # mask-token positions (:266-271, :310-311, :361): every index is shifted by (block_len - 1) at EVERY
                # placeholder and by (S - L - 1) after left padding ("note: -1") -- reproduced as written
                idx = [p for p, t in enumerate(ids_l) if t in mask_ids] if is_avs else list(range(2, 8))
                for sg in plans[i]:
                    if sg[0] != "text":
                        blk = (vfeat[sg[1]] if sg[0] == "video" else afeat[sg[1]]).shape[0]
                        idx = [j + blk - 1 for j in idx]
                idx = [j + S - lens[i] - 1 for j in idx]
                for j in idx:
                    if 0 <= j < S:
                        mtm[i, j] = True
            dict_data['multi_scale_image_features'] = [torch.stack(m, dim=0) for m in ms if len(m) > 0]
            dict_data['mask_token_mask'] = mtm.to(device)
        if return_gt_mask:
            gts = []
            for i in range(bs):
                g = batch_X_modals[i].get('<mask>') if batch_task_names[i] in AVS_TASKS else None
                gts.append(g.to(device).float() if g is not None else torch.zeros((1, 224, 224), device=device))
            dict_data['gt_mask'] = torch.stack(gts, dim=0)
        return dict_data

    def _encode_blocks(self, blocks: Sequence[torch.Tensor], video: bool, want_vit: bool = False):
        """Encode every <video>/<image> (or <audio>) block of the batch in as few launches as possible: blocks of
        equal shape are stacked into one encoder call (the reference encodes them one by one, unified_arch.py:283-300).
        Returns (projector features per block, per-block list of CLIP feature levels when want_vit)."""
        if not blocks:
            return [], []
        out: List[Optional[torch.Tensor]] = [None] * len(blocks)
        vit_out: List[Optional[List[torch.Tensor]]] = [None] * len(blocks)
        groups: Dict[tuple, List[int]] = {}
        for i, b in enumerate(blocks):
            groups.setdefault(tuple(b.shape), []).append(i)
        # at most ENC_CHUNK blocks per encoder call: the towers' scratch is ~60 MB per clip (CLIP: 8 frames x 257 tokens x 29 KB of rows), which at
        # several hundred clips per generate() would take tens of GB away from the KV cache; rows are independent and every chunk is far inside
        # the large-M regime of the kernels, so the features do not depend on the chunking.  The sizes come from plan_enc_chunks (whole tile rounds).
        for shape, all_idxs in groups.items():
            rows = 0
            if video and len(shape) == 4:                        # [frames, 3, H, W]: frames x (patches + CLS) token rows through the CLIP tower
                ve = getattr(self.get_model(), "visual_encoder", None)
                ps = int(ve.vision_tower.config.get("patch_size", 14)) if ve is not None else 14
                rows = shape[0] * ((shape[2] // ps) * (shape[3] // ps) + 1)
            c0 = 0
            for csz in plan_enc_chunks(len(all_idxs), rows, ENC_CHUNK):
                idxs = all_idxs[c0:c0 + csz]
                c0 += csz
                x = torch.stack([blocks[i] for i in idxs], dim=0)
                if video:
                    vit, qf = self.encode_video(x, batch_first=True)
                    f = qf[-1]
                    if want_vit:
                        for j, i in enumerate(idxs):
                            vit_out[i] = [v[j] for v in vit]
                else:
                    f = self.encode_audio(x, batch_first=True)
                for j, i in enumerate(idxs):
                    out[i] = f[j]
        return out, vit_out

    def initialize_MM_tokenizer(self, tokenizer, mask_token_nums=6, output_embeddings_require_grad=False, use_vqgan=False):
        """unified_arch.py:409-459: 11 special + mask_token_nums `<mask_i>` tokens appended in fixed order, tables
        KEYS / MASK / SPECIAL_TOKEN_2_IDS / IDS_2_SPECIAL_TOKEN, then resize_token_embeddings(len(tokenizer))."""
        vocab_nums = len(tokenizer)
        added_tokens = []
        added_tokens += ['<image>', '<image_start>', '<image_end>']
        added_tokens += ['<video>', '<video_start>', '<video_end>']
        added_tokens += ['<audio>', '<audio_start>', '<audio_end>']
        added_tokens += ['<mask_start>', '<mask_end>']
        tokenizer.add_tokens(list(added_tokens), special_tokens=True)
        if use_vqgan:
            # reproduced as shipped (:422-426): the 3 + 16384 VQGAN tokens get ids in the TABLES only - they are never added
            # to the tokenizer, so resize_token_embeddings below does not grow for them - and the KEYS.append('<mask>') of
            # the reference is overwritten by the KEYS assignment further down (the '<mask>' branch of
            # prepare_multimodal_inputs is only reachable if a caller extends KEYS afterwards)
            added_tokens += ['<mask>', '<vqgan_start>', '<vqgan_end>'] + [f'<vqgan_{i}>' for i in range(16384)]
        seg_tokens = [f'<mask_{i}>' for i in range(mask_token_nums)]
        tokenizer.add_tokens(seg_tokens, special_tokens=False)
        added_tokens += seg_tokens
        self.KEYS = ['<image>', '<video>', '<audio>']
        self.MASK = seg_tokens
        self.SPECIAL_TOKEN_2_IDS = {token: i + vocab_nums for i, token in enumerate(added_tokens)}
        self.IDS_2_SPECIAL_TOKEN = {i + vocab_nums: token for i, token in enumerate(added_tokens)}
        self.resize_token_embeddings(len(tokenizer))

    @property
    def device(self):
        return next(self.parameters()).device
