"""`ClassicTextProcessingEngine` over the native CLIP executor -- mirror of backend/text_processing/classic_engine.py for
the arithmetic half: `encode_with_transformers` (:124-148), `process_tokens` (:263-316: BOS / EOS framing of 75-token
chunks is the tokenizer side; here the emphasis multipliers and the "Original" mean restoration of emphasis.py:34-42),
`process_texts`-level chunk concatenation (:252-261 hstack of the per-chunk encodings).

Tokenisation (CLIPTokenizer vocabulary + merges, prompt-attention syntax parsing, locating textual-inversion names in the text) is
host-side string work outside the GPU path: the engine takes token-id / multiplier batches (+ the embedding vectors to splice in), e.g. from the user's Forge install
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
        if emphasis_name not in ("Original", "No norm", "Ignore", "None"):  # backend/text_processing/emphasis.py:19-59
            raise ValueError(f"unknown emphasis mode {emphasis_name}")
        self.emphasis_name = emphasis_name
        self.chunk_length = 75

    def encode_with_transformers(self, tokens, fixes=None):
        """:124-148 -> z [B, 77, C] fp32 with attribute .pooled when return_pooled"""
        layer = max(self.clip_skip, self.minimal_clip_skip)
        z, pooled = self.text_encoder.encode(tokens, clip_skip=layer, final_layer_norm=self.final_layer_norm, return_pooled=self.return_pooled,
                                             project_pooled=self.text_projection and self.embedding_key != "clip_l", fixes=fixes)
        if self.return_pooled:
            z.pooled = pooled
        return z

    def process_tokens(self, remade_batch_tokens, batch_multipliers, fixes=None):
        """:263-316: one 77-token chunk per prompt -> encodings with the emphasis multipliers applied.  fixes: per prompt
        [(offset, vectors), ...] textual-inversion embeddings (the PromptChunkFix entries of :14, with `embedding.vec` already picked
        for this encoder's key)."""
        tokens = torch.as_tensor(remade_batch_tokens)
        z = self.encode_with_transformers(tokens, fixes)
        pooled = getattr(z, "pooled", None)
        if self.emphasis_name in ("Original", "No norm"):  # emphasis.py:34-51; "Ignore" / "None" leave z alone
            m = torch.as_tensor(batch_multipliers, dtype=z.dtype, device=z.device)
            original_mean = z.mean()
            z = z * m.reshape(m.shape + (1,)).expand(z.shape)
            if self.emphasis_name == "Original":
                z = z * (original_mean / z.mean())
        if pooled is not None:
            z.pooled = pooled
        return z

    def __call__(self, chunked_tokens, chunked_multipliers, chunked_fixes=None):
        """chunked_tokens / chunked_multipliers: [n_chunks][B][77] -> [B, 77 * n_chunks, C] (:252-261); .pooled = first chunk's"""
        fx = chunked_fixes if chunked_fixes is not None else [None] * len(chunked_tokens)
        zs = [self.process_tokens(t, m, f) for t, m, f in zip(chunked_tokens, chunked_multipliers, fx)]
        out = torch.hstack(zs)
        if self.return_pooled:
            out.pooled = zs[0].pooled
        return out
