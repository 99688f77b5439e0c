"""HF `transformers` CLIP as an INDEPENDENT second implementation of the tower arithmetic -- TEST INFRASTRUCTURE.

Only `tests/` and `oracle/make_golden.py` import this file.  The reference computes its towers in `open_clip` (reference
models/clip_wrapper.py:13,47,51; models/model_wrapper.py:58,72), which is absent here; `oracle/clip_ref.py` restates it.
HF's CLIP is a separately written implementation of the same published architecture: built from a CONFIG (no hub access)
and filled with the same seeded weights through a q/k/v <-> in_proj key mapping, it pins the restatement -- and, through
`tests/golden/hf_clip_vitb16.npz`, the HIP towers -- to something that shares no code with this repository."""
from __future__ import annotations

from typing import Dict

import torch

from . import clip_ref


def build_hf_clip(cfg: clip_ref.ClipDims, sd: Dict[str, torch.Tensor]):
    """transformers.CLIPModel of `cfg`'s dimensions (eager attention, exact-erf GELU) carrying the open_clip-layout
    state dict `sd`."""
    import transformers

    v, t = cfg.vision, cfg.text
    act = "quick_gelu" if cfg.quick_gelu else "gelu"
    hf_cfg = transformers.CLIPConfig(
        vision_config=dict(hidden_size=v.width, intermediate_size=v.mlp, num_hidden_layers=v.layers,
                           num_attention_heads=v.heads, image_size=cfg.image_size, patch_size=cfg.patch,
                           hidden_act=act, projection_dim=cfg.embed_dim, attn_implementation="eager"),
        text_config=dict(hidden_size=t.width, intermediate_size=t.mlp, num_hidden_layers=t.layers,
                         num_attention_heads=t.heads, vocab_size=cfg.vocab, max_position_embeddings=cfg.ctx,
                         hidden_act=act, projection_dim=cfg.embed_dim, eos_token_id=cfg.vocab - 1,
                         attn_implementation="eager"),
        projection_dim=cfg.embed_dim)
    model = transformers.CLIPModel(hf_cfg).eval()
    model.set_attn_implementation("eager")  # (the sub-configs' request is overridden by the model default otherwise: no attention weights)
    hsd = model.state_dict()

    def put(k, val):
        assert hsd[k].shape == val.shape, (k, hsd[k].shape, val.shape)
        hsd[k] = val.clone()

    def tower(src, dst, layers, d):
        for i in range(layers):
            s, o = f"{src}resblocks.{i}.", f"{dst}.encoder.layers.{i}."
            w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                put(o + f"self_attn.{nm}.weight", w[j * d:(j + 1) * d])
                put(o + f"self_attn.{nm}.bias", b[j * d:(j + 1) * d])
            put(o + "self_attn.out_proj.weight", sd[s + "attn.out_proj.weight"])
            put(o + "self_attn.out_proj.bias", sd[s + "attn.out_proj.bias"])
            for a, bb in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2")):
                put(o + bb + ".weight", sd[s + a + ".weight"])
                put(o + bb + ".bias", sd[s + a + ".bias"])
            for a, bb in (("c_fc", "fc1"), ("c_proj", "fc2")):
                put(o + f"mlp.{bb}.weight", sd[s + f"mlp.{a}.weight"])
                put(o + f"mlp.{bb}.bias", sd[s + f"mlp.{a}.bias"])

    tower("visual.transformer.", "vision_model", v.layers, v.width)
    tower("transformer.", "text_model", t.layers, t.width)
    put("vision_model.embeddings.patch_embedding.weight", sd["visual.conv1.weight"])
    put("vision_model.embeddings.class_embedding", sd["visual.class_embedding"])
    put("vision_model.embeddings.position_embedding.weight", sd["visual.positional_embedding"])
    put("vision_model.pre_layrnorm.weight", sd["visual.ln_pre.weight"])
    put("vision_model.pre_layrnorm.bias", sd["visual.ln_pre.bias"])
    put("vision_model.post_layernorm.weight", sd["visual.ln_post.weight"])
    put("vision_model.post_layernorm.bias", sd["visual.ln_post.bias"])
    put("visual_projection.weight", sd["visual.proj"].t())
    put("text_model.embeddings.token_embedding.weight", sd["token_embedding.weight"])
    put("text_model.embeddings.position_embedding.weight", sd["positional_embedding"])
    put("text_model.final_layer_norm.weight", sd["ln_final.weight"])
    put("text_model.final_layer_norm.bias", sd["ln_final.bias"])
    put("text_projection.weight", sd["text_projection"].t())
    model.load_state_dict(hsd, strict=True)
    return model


@torch.no_grad()
def image_features(model, images: torch.Tensor) -> torch.Tensor:
    out = model.get_image_features(pixel_values=images)
    return getattr(out, "pooler_output", out)


@torch.no_grad()
def text_features(model, tokens: torch.Tensor) -> torch.Tensor:
    out = model.get_text_features(input_ids=tokens)
    return getattr(out, "pooler_output", out)


@torch.no_grad()
def raw_text_transformer(model, x: torch.Tensor):
    """HF's text encoder layers on [n, T, D] as FullModel drives open_clip's `model.transformer` (reference
    models/model_wrapper.py:58,72): no positional embedding, no mask, no final LayerNorm.  Returns (hidden [n,T,D], the
    last layer's per-head softmax probabilities [n,H,T,T])."""
    cap = {}
    last = model.text_model.encoder.layers[-1].self_attn
    h = last.register_forward_hook(lambda m, i, o: cap.update(p=o[1]))
    try:
        hidden = model.text_model.encoder(inputs_embeds=x, attention_mask=None, output_attentions=True).last_hidden_state
    finally:
        h.remove()
    return hidden, cap.get("p")
