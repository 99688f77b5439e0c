"""`FullModel`: image encode -> attribution over the prompt-prefixed text transformer -> prompt
scaling -> text encode -> cosine logits (+ cross-entropy), on the MI355X towers.

Drop-in for reference models/model_wrapper.py:12-100: same constructor, `forward(images,
labels=None) -> {"logits"[, "loss", "loss_cls"]}`, `prompt_learner`, `logit_scale`, state-dict
keys.  What changes is HOW the text side is evaluated:

* collapsed text path (default).  In the reference the text features do not depend on the image:
  per class it runs B identical batch-1 passes for the hook plus one batch-B pass on B identical
  rows (model_wrapper.py:48-75).  That is exactly n_cls sequences x 2 passes and one
  [B,E]x[E,n_cls] product (tests prove literal == collapsed), so this
  module runs 2 text-tower launches per forward instead of n_cls*(B+1).
  `collapse_text=False` replays the reference's loop nest verbatim (slow; parity checks only).
* the attention capture follows the CLIP wrapper's `attn_semantics` ("intended" head-mean softmax
  map vs the "literal" hook output, see clip_wrapper.py here).
* multi-GPU: with `torch.distributed` initialised and `gather_images=True` every rank encodes its
  own image shard and the L2-normalised embeddings are all-gathered (RCCL over xGMI) before the
  logits, so every rank returns logits for the GLOBAL batch (rank-major row order); labels passed to
  `forward` are the rank's LOCAL labels and are gathered the same way, so loss and context gradients are
  those of the global batch on every rank.
"""
import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine
from .attribution_monitor import AttributionMonitor
from .prompt_adjustor import PromptAdjustor
from .prompt_learner import PromptLearner


class _TextTowerFn(torch.autograd.Function):
    """prompts [n,T,D] -> L2-normalised text features [n,E] (reference model_wrapper.py:72-75) with a HIP
    backward: d(pool, projection, norm) and dX through the 12 frozen blocks (`tapclip_text_backward`)."""

    @staticmethod
    def forward(ctx, prompts, clip, tail_run=1):
        tower = clip._text
        # the blocks keep their activations for the backward (no recomputation); like the reference's second
        # pass (model_wrapper.py:72) this call also leaves a capture in clip.attention_maps -- not needed here,
        # so the hook list is left as pass 1 filled it
        hidden, saved = tower.forward_saved(prompts.detach(), tail_run=tail_run)
        ctx.tower = tower
        ctx.tail_run = tail_run
        ctx.save_for_backward(hidden, saved)
        return tower.pool_project(hidden, index=None, ln_final=False, normalize=True)

    @staticmethod
    def backward(ctx, grad_feat):
        hidden, saved = ctx.saved_tensors
        g_hidden = ctx.tower.pool_project_backward(hidden, grad_feat.contiguous(), normalize=True)
        # (tail_run > 1: the gradient of the tied padding rows comes back summed in the run's first row -- those rows
        # are the frozen token bank's, nothing reads them; the context rows get the untied gradients)
        return ctx.tower.backward_saved(saved, g_hidden, tail_run=ctx.tail_run), None, None


class _BuildPromptsFn(torch.autograd.Function):
    """cat([ctx * attribution[..., None], tok], 1) (reference prompt_adjustor.py:35-36, model_wrapper.py:69) as ONE kernel each
    way: with it the training forward issues no torch arithmetic between the towers (`PromptAdjustor('scale')` only; the
    attribution is a constant of the step, as the reference's hook detaches it)."""

    @staticmethod
    def forward(ctx, context, tok, attribution):
        ctx.P = context.shape[1]
        ctx.save_for_backward(attribution)
        return engine.build_prompts(context, tok, attribution)

    @staticmethod
    def backward(ctx, grad_out):
        (attribution,) = ctx.saved_tensors
        return engine.build_prompts_backward(grad_out, ctx.P, attribution), None, None


class _LogitsFn(torch.autograd.Function):
    """exp(logit_scale) * img @ txt.T (reference model_wrapper.py:79,83); img carries no gradient (frozen tower)."""

    @staticmethod
    def forward(ctx, img, txt, log_scale):
        scale = float(log_scale.detach().exp())
        out = engine.logits(img, txt, scale)
        ctx.scale = scale
        ctx.save_for_backward(img, out)
        return out

    @staticmethod
    def backward(ctx, grad_logits):
        img, out = ctx.saved_tensors
        d_txt, d_ls = engine.logits_backward(grad_logits.contiguous(), out, img, ctx.scale)
        return None, d_txt, d_ls


class FullModel(nn.Module):
    def __init__(self, class_names, clip_wrapper, prompt_len=5, attr_lambda=1.0, stab_lambda=0.1,
                 adjustor_method='scale', class_specific=False, *, collapse_text: bool = True,
                 gather_images: bool = False, overlap_towers: bool = True, cache_text_features: bool = False,
                 tie_padding: bool = True):
        super().__init__()
        self.clip = clip_wrapper
        self.class_names = class_names
        self.prompt_learner = PromptLearner(class_names, clip_wrapper, prompt_len, class_specific)
        self.n_cls = len(class_names)
        self.attribution_monitor = AttributionMonitor(prompt_len)
        self.prompt_adjustor = PromptAdjustor(method=adjustor_method)
        # stored, never used by forward -- as in the reference (model_wrapper.py:24-25)
        self.attr_lambda = attr_lambda
        self.stab_lambda = stab_lambda
        # log(1/0.07): exp() = 14.2857, NOT the pretrained CLIP scale (model_wrapper.py:26)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.collapse_text = collapse_text
        self.gather_images = gather_images
        # gather_images with UNEQUAL local batches (the short last batch of an evaluation loader sharded without padding):
        # row counts are exchanged first and the shards padded for the collective (dist.all_gather_rows(ragged=True)).
        # Off in the hot path -- evenly sharded batches need no extra collective; the evaluation helpers of
        # utils/eval_metrics.py switch it on for their loops.
        self.ragged_batches = False
        # The image tower does not depend on the text side: it is launched on a second HIP stream so that its
        # chip-filling kernels run beside the text tower's small grids (65 x 93 rows: 96-192 workgroups per GEMM
        # on 256 CUs) instead of after them.
        self.overlap_towers = overlap_towers
        self._side_stream = None
        # Evaluation helper (off by default; bench.py never uses it): the text features do not depend on the images,
        # so a no-grad forward may re-use them while no prompt parameter, CLIP weight or adjustor setting has changed.
        # The reference recomputes them for every batch (model_wrapper.py:47-75).
        self.cache_text_features = cache_text_features
        self._text_cache = None
        # The padding positions of every class prompt carry ONE embedding row (zero-padded token ids, reference
        # prompt_learner.py:31-33) and the transformer is fed without position or mask (model_wrapper.py:58,72): those rows
        # stay identical through every block, so the collapsed text path merges them -- 26 instead of 93 rows per sequence
        # at BASELINE configs[2], same logits, maps and gradients (include/tapclip.h "tied padding rows").  False runs
        # every row, as the literal replay always does.
        self.tie_padding = tie_padding and os.environ.get("TAPCLIP_TIE_PADDING", "1") != "0"  # (the variable: A/B runs of tools/)
        # a checkpoint that carries other `clip.model.*` weights re-packs the towers (clip_wrapper.py here);
        # the frozen class-token embeddings derived from them are then re-computed as well
        self._clip_version = getattr(clip_wrapper, "weights_version", 0)
        self.register_load_state_dict_post_hook(FullModel._refresh_after_load)

    @staticmethod
    def _refresh_after_load(module, incompatible_keys):
        v = getattr(module.clip, "weights_version", 0)
        if v != module._clip_version:
            module._clip_version = v
            module.prompt_learner.refresh_token_bank()

    # ---- scheduling of the towers' GEMMs ---------------------------------------------------------
    class _ForwardGemms:
        """Context of every FORWARD pass FullModel drives: both towers launch their GEMMs without the K-split of partial
        rounds (`TAPCLIP_FLAG_KSPLIT` 0).  The split buys latency with CU-time -- right for a tower alone on the GPU, wrong
        while the other tower's stream can use the idle CUs: the overlapped forward of BASELINE configs[2] is 12.7 ms with
        the splits and 12.2 ms without (tools/train_phases.py, same box, interleaved).  The serial order
        (`overlap_towers=False`) runs under the same setting, so the two orders stay bit-identical; the backward runs
        after the context, split again (the text tower then has the GPU to itself)."""

        def __init__(self, clip):
            self.clip = clip

        def __enter__(self):
            clip = self.clip
            if getattr(clip, "_forward_gemms_depth", 0) == 0:
                # what each tower was set to BEFORE this forward (a caller may have chosen 0 for a tower that shares the
                # GPU): that, not the library default, is what __exit__ puts back.  The depth is only bumped once both
                # flags are set, so a failing call leaves nothing half-entered.
                towers = (clip._vision, clip._text)
                before = [t.get_ksplit() for t in towers]
                done = []
                try:
                    for t in towers:
                        t.set_ksplit(False)
                        done.append(t)
                except Exception:
                    for t, v in zip(done, before):
                        t.set_ksplit(v)
                    raise
                clip._forward_gemms_saved = list(zip(towers, before))
            clip._forward_gemms_depth = getattr(clip, "_forward_gemms_depth", 0) + 1

        def __exit__(self, *exc):
            clip = self.clip
            clip._forward_gemms_depth -= 1
            if clip._forward_gemms_depth == 0:
                live = (clip._vision, clip._text)  # (the towers may have been re-packed meanwhile: address them anew)
                for (old, v), t in zip(clip._forward_gemms_saved, live):
                    t.set_ksplit(v)
                clip._forward_gemms_saved = []
            return False

    # ---- image side ----------------------------------------------------------------------------
    def _image_features_begin(self, images: torch.Tensor, labels=None):
        """Launch `encode_image` (reference model_wrapper.py:40-41) and, in the data-parallel form, the one exchange step
        behind it; returns (features, labels, stream to join).  With `overlap_towers` both run on a second HIP stream: the
        image tower beside the text tower's small grids, and the all-gather of the embeddings (RCCL over xGMI, 512 KiB per
        rank) beside whatever the replicated text tower still has to do -- the main stream only waits for it at the logits."""
        vision = self.clip._vision
        if not (self.overlap_towers and images.is_cuda):
            feat = vision.encode_image(images, normalize=True)
            feat, labels = self._gather(feat, labels)
            return feat, labels, None
        dev = images.device
        if self._side_stream is None:
            # (TAPCLIP_IMAGE_STREAM_PRIORITY: experiments -- -1 asks for a high-priority stream; measured twice, rounds 3 and 4: no effect)
            prio = int(os.environ.get("TAPCLIP_IMAGE_STREAM_PRIORITY", "0"))
            self._side_stream = torch.cuda.Stream(device=dev, priority=prio)
        side = self._side_stream
        side.wait_stream(torch.cuda.current_stream(dev))  # the images (and labels) were produced on the caller's stream
        with torch.cuda.stream(side):
            feat = vision.encode_image(images, normalize=True)
            feat, labels = self._gather(feat, labels)
        return feat, labels, side

    @staticmethod
    def _image_features_end(feat: torch.Tensor, labels, side):
        if side is not None:
            main = torch.cuda.current_stream(feat.device)
            main.wait_stream(side)
            feat.record_stream(main)
            if labels is not None and labels.is_cuda:
                labels.record_stream(main)
        return feat, labels

    # ---- text side -----------------------------------------------------------------------------
    def _text_cache_key(self):
        params = list(self.prompt_learner.parameters()) + list(self.prompt_adjustor.parameters())
        return (tuple((p.data_ptr(), p._version) for p in params), getattr(self.clip, "weights_version", 0),
                self.prompt_adjustor.method, self.clip.attn_semantics, len(self.prompt_learner.context_bank))

    def text_features(self) -> torch.Tensor:
        """L2-normalised text features [n_cls, E] (reference model_wrapper.py:47-75, collapsed)."""
        if self.cache_text_features and not torch.is_grad_enabled():
            key = self._text_cache_key()
            if self._text_cache is not None and self._text_cache[0] == key:
                return self._text_cache[1]
            feats = self._text_features_uncached()
            self._text_cache = (key, feats)
            return feats
        return self._text_features_uncached()

    def _text_features_uncached(self) -> torch.Tensor:
        with FullModel._ForwardGemms(self.clip):  # (re-entrant: forward() is already inside one)
            return self._text_features_passes()

    def _text_features_passes(self) -> torch.Tensor:
        pl, clip = self.prompt_learner, self.clip
        P = pl.prompt_len
        ctx, tok = pl.stacked_context().detach(), pl.stacked_tokens()
        fused = self.prompt_adjustor.method == "scale"

        run = self._tail_run()
        # pass 1: only for the attention capture (model_wrapper.py:57-62)
        clip.reset()
        clip.model.transformer.capture(engine.build_prompts(ctx, tok), _tail_run=run)
        attn_map = clip.get_attention_map()
        if attn_map.dim() == 2:
            attn_map = attn_map.unsqueeze(1)  # per sample [1,D] -> [1,1,D] in the reference (:60-61)
        attribution = self.attribution_monitor(attn_map)  # [n_cls, P] (or [n_cls, 1] literal)

        # pass 2: adjusted prompt -> last token -> projection -> norm (model_wrapper.py:68-75)
        if fused:
            adjusted = engine.build_prompts(ctx, tok, attribution)
        elif not torch.is_grad_enabled() and ctx.is_cuda:
            adjusted = engine.build_prompts_mlp(ctx, tok, attribution, self.prompt_adjustor)   # 'gate' / 'residual', one kernel
        else:
            adjusted = torch.cat([self.prompt_adjustor(ctx, attribution), tok], dim=1)
        hidden = clip.model.transformer(adjusted, _tail_run=run)
        self.last_attribution = attribution
        return clip._text.pool_project(hidden, index=None, ln_final=False, normalize=True)

    def _tail_run(self) -> int:
        return self.prompt_learner.tail_run() if self.tie_padding else 1

    def check_tied_padding(self) -> None:
        """Raise if the text tower found the tied-padding claim false (the rows FullModel said were identical were not: a
        direct edit of `prompt_learner.token_bank`, a user hook that changes rows in front of the transformer).  The library
        then poisons its outputs with NaN -- safe, but a NaN loss does not say why.  One device-to-host read: call it where
        the host synchronises anyway (the evaluation loops do at their end; `forward` does when its loss is not finite).
        The cached run length is dropped, so the next forward measures the token bank again."""
        if not self.tie_padding or not hasattr(self.clip, "_text"):
            return
        if self.clip._text.tied_violations():
            self.prompt_learner._tail_run = None
            self.prompt_learner._tok_cache = None
            raise RuntimeError("tied padding rows: the text tower was told that the last `tail_run` rows of every prompt are identical "
                               "(FullModel(tie_padding=True), PromptLearner.tail_run()) and found them different -- its outputs since "
                               "then are NaN on purpose.  Rebuild the token bank with PromptLearner.refresh_token_bank() after editing "
                               "it, or construct FullModel(tie_padding=False) / set TAPCLIP_TIE_PADDING=0")

    def _forward_literal(self, images: torch.Tensor) -> torch.Tensor:
        """The reference loop nest as written (model_wrapper.py:47-83), on the HIP towers."""
        pl, clip = self.prompt_learner, self.clip
        P = pl.prompt_len
        B = images.size(0)
        raw = pl().detach()
        image_feat = clip._vision.encode_image(images, normalize=True)
        scale = float(self.logit_scale.detach().exp())
        sims = []
        for i, _name in enumerate(pl.context_bank.keys()):
            ctx = raw[i, :P].unsqueeze(0).expand(B, -1, -1).contiguous()
            cls_tok = raw[i, P:].unsqueeze(0).expand(B, -1, -1).contiguous()
            attrs = []
            for b in range(B):
                clip.reset()
                clip.model.transformer(engine.build_prompts(ctx[b: b + 1], cls_tok[b: b + 1]))
                amap = clip.get_attention_map()
                if amap.dim() == 2:
                    amap = amap.unsqueeze(0)
                attrs.append(self.attribution_monitor(amap))
            attribution = torch.cat(attrs, dim=0)
            hidden = clip.model.transformer(engine.build_prompts(ctx, cls_tok, attribution))
            tf = clip._text.pool_project(hidden, normalize=True)                 # [B, E], rows identical
            sims.append(scale * (image_feat * tf).sum(dim=-1, keepdim=True))
        return torch.cat(sims, dim=1)

    # ---- forward -------------------------------------------------------------------------------
    def _forward_train(self, images, labels):
        """Differentiable forward (reference train.py:99-105): gradients reach `context_bank.*` (and any
        trainable PromptAdjustor net) and `logit_scale`; the attention capture is a constant, as the
        reference's hook detaches it (clip_wrapper.py:36)."""
        pl, clip = self.prompt_learner, self.clip
        with FullModel._ForwardGemms(clip):  # (the backward, later, runs with the K-split on again)
            with torch.no_grad():
                image_feat, labels, side = self._image_features_begin(images, labels)
                ctx_c, tok = pl.stacked_context().detach(), pl.stacked_tokens()
                run = self._tail_run()
                clip.reset()
                clip.model.transformer.capture(engine.build_prompts(ctx_c, tok), _tail_run=run)
                attn_map = clip.get_attention_map()
                if attn_map.dim() == 2:
                    attn_map = attn_map.unsqueeze(1)
                attribution = self.attribution_monitor(attn_map)
            ctx = pl.stacked_context()                                   # differentiable w.r.t. every context_bank entry
            if self.prompt_adjustor.method == "scale" and ctx.is_cuda:
                adjusted = _BuildPromptsFn.apply(ctx, tok, attribution)
            else:  # 'gate' / 'residual' carry trainable nets of their own: the torch modules and their autograd
                adjusted = torch.cat([self.prompt_adjustor(ctx, attribution), tok], dim=1)
            text_feat = _TextTowerFn.apply(adjusted, clip, run)
        with torch.no_grad():
            image_feat, labels = self._image_features_end(image_feat, labels, side)
        logits = _LogitsFn.apply(image_feat, text_feat, self.logit_scale)
        self.last_attribution = attribution
        outputs = {"logits": logits}
        if labels is not None:
            loss_cls = F.cross_entropy(logits, labels)
            outputs.update({"loss": loss_cls, "loss_cls": loss_cls})
        return outputs

    def _gather(self, image_feat: torch.Tensor, labels):
        """The one exchange step of the data-parallel path: every rank contributes its
        [B_local, E] embeddings and its [B_local] labels, in the same rank-major order, so logits, loss and the
        context gradients are those of the GLOBAL batch and identical on every rank (no gradient all-reduce)."""
        if labels is not None:
            labels = labels.to(image_feat.device)
        if self.gather_images:
            from ..dist import all_gather_rows
            image_feat = all_gather_rows(image_feat, ragged=self.ragged_batches)
            if labels is not None:
                labels = all_gather_rows(labels, ragged=self.ragged_batches)
        return image_feat, labels

    def forward(self, images, labels=None):
        wants_grad = torch.is_grad_enabled() and (
            self.logit_scale.requires_grad or any(p.requires_grad for p in self.prompt_learner.parameters()))
        if wants_grad and self.collapse_text:
            return self._forward_train(images, labels)
        if wants_grad and labels is not None:
            # the literal replay exists for parity checks only and runs without autograd: a loss from it could not
            # be differentiated (the reference's own loop can, at n_cls * (B + 1) text passes per step)
            raise RuntimeError("FullModel(collapse_text=False) has no backward: build the model with collapse_text=True "
                               "to train, or call it under torch.no_grad() for the literal replay")
        with torch.no_grad():
            if not self.collapse_text:
                if self.gather_images:
                    raise RuntimeError("FullModel(collapse_text=False) is single-process (parity replay of the reference loop)")
                with FullModel._ForwardGemms(self.clip):  # (same summation order as the collapsed forward)
                    logits = self._forward_literal(images)
                labels = None if labels is None else labels.to(logits.device)
            else:
                with FullModel._ForwardGemms(self.clip):
                    image_feat, labels, side = self._image_features_begin(images, labels)  # model_wrapper.py:40-41
                    text_feat = self.text_features()
                image_feat, labels = self._image_features_end(image_feat, labels, side)
                logits = engine.logits(image_feat, text_feat, float(self.logit_scale.exp()))  # :79,83
            outputs = {"logits": logits}
            if labels is not None:
                loss_cls = F.cross_entropy(logits, labels)
                outputs.update({"loss": loss_cls, "loss_cls": loss_cls})
        return outputs
