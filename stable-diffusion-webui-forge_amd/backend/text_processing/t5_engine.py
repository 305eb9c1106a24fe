"""T5 text-processing engine -- mirror of backend/text_processing/t5_engine.py:18-158 (`T5TextProcessingEngine`): prompt -> emphasis-parsed pieces ->
T5 token chunks (EOS = 1 appended, padded with 0 to min_length 256; BREAK starts a new chunk) -> encoder -> emphasis multipliers applied to the
encodings.  The tokenizer (a transformers T5TokenizerFast: its sentencepiece vocabulary is a data file) is handed in; the encoder is the native
`backend.nn.t5.IntegratedT5`."""
from collections import namedtuple

import torch

from . import parsing

PromptChunkFix = namedtuple("PromptChunkFix", ["offset", "embedding"])


class PromptChunk:
    def __init__(self):
        self.tokens = []
        self.multipliers = []


class T5TextProcessingEngine:
    def __init__(self, text_encoder, tokenizer, emphasis_name="Original", min_length=256):
        self.text_encoder = text_encoder.transformer
        self.tokenizer = tokenizer
        if emphasis_name not in ("Original", "No norm", "Ignore", "None"):   # backend/text_processing/emphasis.py:19-59
            raise ValueError(f"unknown emphasis mode {emphasis_name}")
        self.emphasis_name = emphasis_name
        self.min_length = min_length
        self.id_end = 1
        self.id_pad = 0
        vocab = self.tokenizer.get_vocab()
        self.comma_token = vocab.get(",</w>", None)
        self.token_mults = {}
        for text, ident in [(k, v) for k, v in vocab.items() if "(" in k or ")" in k or "[" in k or "]" in k]:   # :38-53
            mult = 1.0
            for c in text:
                if c == "[":
                    mult /= 1.1
                if c == "]":
                    mult *= 1.1
                if c == "(":
                    mult *= 1.1
                if c == ")":
                    mult /= 1.1
            if mult != 1.0:
                self.token_mults[ident] = mult

    def tokenize(self, texts):
        return self.tokenizer(texts, truncation=False, add_special_tokens=False)["input_ids"]

    def encode_with_transformers(self, tokens):
        return self.text_encoder(input_ids=tokens)

    def tokenize_line(self, line):
        """:68-112"""
        parsed = parsing.parse_prompt_attention(line, self.emphasis_name)
        tokenized = self.tokenize([text for text, _ in parsed])
        chunks = []
        chunk = PromptChunk()
        token_count = 0

        def next_chunk():
            nonlocal token_count, chunk
            chunk.tokens = chunk.tokens + [self.id_end]
            chunk.multipliers = chunk.multipliers + [1.0]
            current = len(chunk.tokens)
            token_count += current
            remaining = self.min_length - current
            if remaining > 0:
                chunk.tokens += [self.id_pad] * remaining
                chunk.multipliers += [1.0] * remaining
            chunks.append(chunk)
            chunk = PromptChunk()

        for tokens, (text, weight) in zip(tokenized, parsed):
            if text == "BREAK" and weight == -1:
                next_chunk()
                continue
            for token in tokens:
                chunk.tokens.append(token)
                chunk.multipliers.append(weight)
        if chunk.tokens or not chunks:
            next_chunk()
        return chunks, token_count

    def __call__(self, texts):
        """:114-145: one encoding per chunk of every line, the chunks of a line padded to its longest; stacked over (line, chunk)"""
        zs = []
        cache = {}
        for line in texts:
            if line in cache:
                line_z = cache[line]
            else:
                chunks, _ = self.tokenize_line(line)
                max_tokens = max(len(c.tokens) for c in chunks)
                line_z = []
                for c in chunks:
                    pad = max_tokens - len(c.tokens)
                    tokens, mults = c.tokens + [self.id_pad] * pad, c.multipliers + [1.0] * pad
                    line_z.append(self.process_tokens([tokens], [mults])[0])
                cache[line] = line_z
            zs.extend(line_z)
        return torch.stack(zs)

    def process_tokens(self, batch_tokens, batch_multipliers):
        """:147-158 + emphasis.py:34-51"""
        z = self.encode_with_transformers(torch.asarray(batch_tokens))
        if self.emphasis_name in ("Original", "No norm"):
            m = torch.asarray(batch_multipliers).to(z)
            if self.emphasis_name == "Original":
                original_mean = z.mean()
                z = z * m.reshape(m.shape + (1,)).expand(z.shape)
                z = z * (original_mean / z.mean())
            else:
                z = z * m.reshape(m.shape + (1,)).expand(z.shape)
        return z
