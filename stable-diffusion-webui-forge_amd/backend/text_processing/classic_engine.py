"""`ClassicTextProcessingEngine` over the native CLIP executor -- mirror of backend/text_processing/classic_engine.py for
the arithmetic half: `encode_with_transformers` (:124-148), `process_tokens` (:263-316: BOS / EOS framing of 75-token
chunks is the tokenizer side; here the emphasis multipliers and the "Original" mean restoration of emphasis.py:34-42),
`process_texts`-level chunk concatenation (:252-261 hstack of the per-chunk encodings).

Tokenisation (CLIPTokenizer vocabulary + merges, prompt-attention syntax parsing, locating textual-inversion names in the text) is
host-side string work outside the GPU path: the engine takes token-id / multiplier batches (+ the embedding vectors to splice in), e.g. from the user's Forge install
(`tokenizer(texts)["input_ids"]`)."""
import math
from collections import namedtuple

import torch

from . import parsing

PromptChunkFix = namedtuple("PromptChunkFix", ["offset", "embedding"])  # classic_engine.py:11


class PromptChunk:
    """classic_engine.py:14-18: 75 prompt tokens framed by BOS / EOS, their emphasis multipliers, textual-inversion fixes."""

    def __init__(self):
        self.tokens, self.multipliers, self.fixes = [], [], []


class ClassicTextProcessingEngine:
    def __init__(self, text_encoder, embedding_key="clip_l", text_projection=False, minimal_clip_skip=1, clip_skip=1,
                 return_pooled=False, final_layer_norm=True, emphasis_name="Original", tokenizer=None, embeddings=None, chunk_length=75):
        """tokenizer: a CLIPTokenizer-like object (`tokenizer(texts, truncation=False, add_special_tokens=False)["input_ids"]`, bos / eos / pad
        token ids, `get_vocab()`), needed only for string prompts (`encode_texts`); token batches work without one.
        embeddings: optional textual-inversion database with `find_embedding_at_position(tokens, position) -> (embedding, n_tokens)`,
        `embedding.vectors`, `embedding.vec` (tensor or {key: tensor})."""
        self.tokenizer, self.embeddings = tokenizer, embeddings
        if tokenizer is not None:
            self.id_start, self.id_end, self.id_pad = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
            self.comma_token = tokenizer.get_vocab().get(",</w>", None)
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
        self.chunk_length = chunk_length

    # ---- string side (classic_engine.py:112-122, 150-250): prompt -> 75-token chunks -------------------------------------------------
    def empty_chunk(self):
        chunk = PromptChunk()
        chunk.tokens = [self.id_start] + [self.id_end] * (self.chunk_length + 1)
        chunk.multipliers = [1.0] * (self.chunk_length + 2)
        return chunk

    def get_target_prompt_token_count(self, token_count):
        return math.ceil(max(token_count, 1) / self.chunk_length) * self.chunk_length

    def tokenize(self, texts):
        if self.tokenizer is None:
            raise RuntimeError("string prompts need a tokenizer (ClassicTextProcessingEngine(..., tokenizer=...)); token batches do not")
        return self.tokenizer(texts, truncation=False, add_special_tokens=False)["input_ids"]

    def tokenize_line(self, line, comma_padding_backtrack=20):
        """-> (chunks, token_count).  Emphasis syntax is parsed first and each run tokenised on its own; tokens fill 75-slot chunks; a chunk
        that fills up within `comma_padding_backtrack` tokens after a comma is cut at that comma instead (the tail moves to the next chunk);
        BREAK closes the chunk; a textual-inversion embedding takes `vectors` placeholder slots and never straddles a chunk boundary."""
        parsed = parsing.parse_prompt_attention(line, self.emphasis_name)
        tokenized = self.tokenize([text for text, _ in parsed])
        chunks, state = [], {"chunk": PromptChunk(), "count": 0, "last_comma": -1}

        def close(is_last=False):
            chunk = state["chunk"]
            state["count"] += len(chunk.tokens) if is_last else self.chunk_length
            pad = self.chunk_length - len(chunk.tokens)
            if pad > 0:
                chunk.tokens += [self.id_end] * pad
                chunk.multipliers += [1.0] * pad
            chunk.tokens = [self.id_start] + chunk.tokens + [self.id_end]
            chunk.multipliers = [1.0] + chunk.multipliers + [1.0]
            chunks.append(chunk)
            state["chunk"], state["last_comma"] = PromptChunk(), -1

        for tokens, (text, weight) in zip(tokenized, parsed):
            if text == "BREAK" and weight == -1:
                close()
                continue
            position = 0
            while position < len(tokens):
                token, chunk = tokens[position], state["chunk"]
                if token == self.comma_token:
                    state["last_comma"] = len(chunk.tokens)
                elif (comma_padding_backtrack != 0 and len(chunk.tokens) == self.chunk_length and state["last_comma"] != -1
                      and len(chunk.tokens) - state["last_comma"] <= comma_padding_backtrack):
                    cut = state["last_comma"] + 1
                    moved_tokens, moved_mults = chunk.tokens[cut:], chunk.multipliers[cut:]
                    chunk.tokens, chunk.multipliers = chunk.tokens[:cut], chunk.multipliers[:cut]
                    close()
                    state["chunk"].tokens, state["chunk"].multipliers = moved_tokens, moved_mults
                if len(state["chunk"].tokens) == self.chunk_length:
                    close()
                chunk = state["chunk"]
                embedding, n_src = (None, None) if self.embeddings is None else self.embeddings.find_embedding_at_position(tokens, position)
                if embedding is None:
                    chunk.tokens.append(token)
                    chunk.multipliers.append(weight)
                    position += 1
                    continue
                n_vec = int(embedding.vectors)
                if len(chunk.tokens) + n_vec > self.chunk_length:
                    close()
                    chunk = state["chunk"]
                chunk.fixes.append(PromptChunkFix(len(chunk.tokens), embedding))
                chunk.tokens += [0] * n_vec
                chunk.multipliers += [weight] * n_vec
                position += n_src
        if state["chunk"].tokens or not chunks:
            close(is_last=True)
        return chunks, state["count"]

    def process_texts(self, texts):
        token_count, cache, batch_chunks = 0, {}, []
        for line in texts:
            if line not in cache:
                cache[line], n = self.tokenize_line(line)
                token_count = max(n, token_count)
            batch_chunks.append(cache[line])
        return batch_chunks, token_count

    def encode_texts(self, texts):
        """classic_engine.py:252-261 `__call__(texts)`: prompts -> [B, 77 * n_chunks, C]; prompts with fewer chunks are padded with empty ones."""
        batch_chunks, _ = self.process_texts(texts)
        n_chunks = max(len(x) for x in batch_chunks)
        toks, mults, fixes = [], [], []
        for i in range(n_chunks):
            row = [chunks[i] if i < len(chunks) else self.empty_chunk() for chunks in batch_chunks]
            toks.append([c.tokens for c in row])
            mults.append([c.multipliers for c in row])
            fixes.append([[(f.offset, f.embedding.vec[self.embedding_key] if isinstance(f.embedding.vec, dict) else f.embedding.vec) for f in c.fixes]
                          for c in row])
        return self(toks, mults, fixes if any(any(f) for f in fixes) else None)

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
