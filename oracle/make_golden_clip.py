"""ORACLE -- pins the CLIP text tower against an INDEPENDENT third-party implementation:
`transformers.CLIPTextModelWithProjection` (transformers 5.15 ships it; openai/CLIP itself is an un-vendored, unpinned git
dependency of the reference -- requirements.txt:7 -- and is not installed anywhere we can run).

The reference calls `clip.load("ViT-B/32")` and `model.clip_model.encode_text(tokens)` (models/dreamvla_model.py:643-650).
The pretrained checkpoint is not available offline, so the pin is architectural: FULL-SIZE ViT-B/32 text tower (12 layers,
width 512, 8 heads, 77 tokens, vocabulary 49408, QuickGELU, causal mask, EOT pooling by arg-max token id, projection 512),
weights from the deterministic recipe of oracle/weights.py under the openai/CLIP state_dict key names (what a real checkpoint
would carry), mapped into the Hugging Face module by the published key correspondence (q/k/v_proj <- in_proj split,
fc1/fc2 <- c_fc/c_proj, text_projection transposed).  Stored: the Hugging Face outputs only (tests/golden/clip_text_hf.pt).

    python -m oracle.make_golden_clip
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
W, LAYERS, HEADS, CTX, VOCAB, PROJ = 512, 12, 8, 77, 49408, 512


def openai_keys():
    keys = {"token_embedding.weight": (VOCAB, W), "positional_embedding": (CTX, W), "ln_final.weight": (W,), "ln_final.bias": (W,),
            "text_projection": (W, PROJ)}
    for i in range(LAYERS):
        b = f"transformer.resblocks.{i}."
        keys.update({b + "attn.in_proj_weight": (3 * W, W), b + "attn.in_proj_bias": (3 * W,), b + "attn.out_proj.weight": (W, W),
                     b + "attn.out_proj.bias": (W,), b + "ln_1.weight": (W,), b + "ln_1.bias": (W,), b + "ln_2.weight": (W,),
                     b + "ln_2.bias": (W,), b + "mlp.c_fc.weight": (4 * W, W), b + "mlp.c_fc.bias": (4 * W,),
                     b + "mlp.c_proj.weight": (W, 4 * W), b + "mlp.c_proj.bias": (W,)})
    return keys


def openai_state_dict():
    """the text tower's tensors under openai/CLIP's names, from the key-seeded recipe (bf16-representable fp32)"""
    return {k: weights.recipe_tensor("clip_model." + k, shp) for k, shp in openai_keys().items()}


def to_hf(sd):
    out = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
           "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
           "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
           "text_projection.weight": sd["text_projection"].t().contiguous()}
    for i in range(LAYERS):
        b, h = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        wq, wk, wv = sd[b + "attn.in_proj_weight"].chunk(3, dim=0)
        bq, bk, bv = sd[b + "attn.in_proj_bias"].chunk(3, dim=0)
        out.update({h + "self_attn.q_proj.weight": wq, h + "self_attn.k_proj.weight": wk, h + "self_attn.v_proj.weight": wv,
                    h + "self_attn.q_proj.bias": bq, h + "self_attn.k_proj.bias": bk, h + "self_attn.v_proj.bias": bv,
                    h + "self_attn.out_proj.weight": sd[b + "attn.out_proj.weight"], h + "self_attn.out_proj.bias": sd[b + "attn.out_proj.bias"],
                    h + "layer_norm1.weight": sd[b + "ln_1.weight"], h + "layer_norm1.bias": sd[b + "ln_1.bias"],
                    h + "layer_norm2.weight": sd[b + "ln_2.weight"], h + "layer_norm2.bias": sd[b + "ln_2.bias"],
                    h + "mlp.fc1.weight": sd[b + "mlp.c_fc.weight"], h + "mlp.fc1.bias": sd[b + "mlp.c_fc.bias"],
                    h + "mlp.fc2.weight": sd[b + "mlp.c_proj.weight"], h + "mlp.fc2.bias": sd[b + "mlp.c_proj.bias"]})
    return out


def tokens(n=6, seed=11):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(1, 49000, (n, CTX), generator=g)
    t[:, 0] = 49406                                     # <|startoftext|>
    eot = torch.randint(4, CTX, (n,), generator=g)
    eot[0] = CTX - 1                                    # a full-length prompt
    for i in range(n):
        t[i, eot[i]] = 49407                            # <|endoftext|> = the largest id: arg-max pooling position
        t[i, eot[i] + 1:] = 0
    return t


def main():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    import transformers
    cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=W, intermediate_size=4 * W, projection_dim=PROJ, num_hidden_layers=LAYERS,
                         num_attention_heads=HEADS, max_position_embeddings=CTX, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                         eos_token_id=49407, bos_token_id=49406, pad_token_id=0)
    hf = CLIPTextModelWithProjection(cfg).eval()
    missing, unexpected = hf.load_state_dict(to_hf(openai_state_dict()), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    tok = tokens()
    with torch.no_grad():
        o = hf(input_ids=tok)
    flat = o.last_hidden_state.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 4096).long()
    torch.save(dict(tokens=tok, text_embeds=o.text_embeds.clone(), hidden_idx=idx, hidden_vals=flat[idx].clone(),
                    hidden_shape=list(o.last_hidden_state.shape),
                    source=f"transformers {transformers.__version__} CLIPTextModelWithProjection, weights = oracle/weights.py recipe "
                           f"under openai/CLIP key names (oracle/make_golden_clip.py)"),
               os.path.join(GOLD, "clip_text_hf.pt"))
    print("clip_text_hf.pt", os.path.getsize(os.path.join(GOLD, "clip_text_hf.pt")), float(o.text_embeds.abs().mean()))


if __name__ == "__main__":
    main()
