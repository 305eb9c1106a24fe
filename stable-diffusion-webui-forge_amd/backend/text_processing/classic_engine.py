"""`ClassicTextProcessingEngine` over the native CLIP executor -- mirror of backend/text_processing/classic_engine.py for
the arithmetic half: `encode_with_transformers` (:124-148), `process_tokens` (:263-316: BOS / EOS framing of 75-token
chunks is the tokenizer side; here the emphasis multipliers and the "Original" mean restoration of emphasis.py:34-42),
`process_texts`-level chunk concatenation (:252-261 hstack of the per-chunk encodings).

Tokenisation (CLIPTokenizer vocabulary + merges, prompt-attention syntax parsing, textual-inversion embeddings) is host-side
string work outside the GPU path: the engine takes token-id / multiplier batches, e.g. from the user's Forge install
(`tokenizer(texts)["input_ids"]`)."""
import torch


class ClassicTextProcessingEngine:
    def __init__(self, text_encoder, embedding_key="clip_l", text_projection=False, minimal_clip_skip=1, clip_skip=1,
                 return_pooled=False, final_layer_norm=True, emphasis_name="Original"):
        self.text_encoder = text_encoder            # forge_amd.backend.nn.clip.IntegratedCLIP
        self.embedding_key = embedding_key
        self.text_projection = text_projection
        self.minimal_clip_skip = minimal_clip_skip
        self.clip_skip = clip_skip
        self.return_pooled = return_pooled
        self.final_layer_norm = final_layer_norm
        if emphasis_name not in ("Original", "None"):
            raise NotImplementedError("emphasis modes other than Original / None")
        self.emphasis_name = emphasis_name
        self.chunk_length = 75

    def encode_with_transformers(self, tokens):
        """:124-148 -> z [B, 77, C] fp32 with attribute .pooled when return_pooled"""
        layer = max(self.clip_skip, self.minimal_clip_skip)
        z, pooled = self.text_encoder.encode(tokens, clip_skip=layer, final_layer_norm=self.final_layer_norm, return_pooled=self.return_pooled,
                                             project_pooled=self.text_projection and self.embedding_key != "clip_l")
        if self.return_pooled:
            z.pooled = pooled
        return z

    def process_tokens(self, remade_batch_tokens, batch_multipliers):
        """:263-316: one 77-token chunk per prompt -> encodings with the emphasis multipliers applied"""
        tokens = torch.as_tensor(remade_batch_tokens)
        z = self.encode_with_transformers(tokens)
        pooled = getattr(z, "pooled", None)
        if self.emphasis_name == "Original":
            m = torch.as_tensor(batch_multipliers, dtype=z.dtype, device=z.device)
            original_mean = z.mean()
            z = z * m.reshape(m.shape + (1,)).expand(z.shape)
            z = z * (original_mean / z.mean())
        if pooled is not None:
            z.pooled = pooled
        return z

    def __call__(self, chunked_tokens, chunked_multipliers):
        """chunked_tokens / chunked_multipliers: [n_chunks][B][77] -> [B, 77 * n_chunks, C] (:252-261); .pooled = first chunk's"""
        zs = [self.process_tokens(t, m) for t, m in zip(chunked_tokens, chunked_multipliers)]
        out = torch.hstack(zs)
        if self.return_pooled:
            out.pooled = zs[0].pooled
        return out
