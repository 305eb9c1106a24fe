"""T5 text processing for Flux: prompt strings -> padded T5 token rows -> `backend.nn.t5.IntegratedT5` -> emphasis-weighted encodings.

Behavioural contract (what a Forge caller observes; reference: backend/text_processing/t5_engine.py:18-158, emphasis.py:19-59):
  * a prompt is cut at emphasis boundaries (`parsing.parse_prompt_attention`), every piece tokenised WITHOUT special tokens, the pieces' tokens
    concatenated with the piece's weight beside each token;
  * the keyword BREAK closes the current row; every row ends with EOS (id 1, weight 1) and is padded with id 0 to `min_length` (256);
  * the rows of one prompt are padded to that prompt's longest row, encoded one by one, and stacked over (prompt, row); a repeated prompt is
    encoded once;
  * emphasis: "Original" multiplies token encodings by their weights and restores the tensor's mean, "No norm" only multiplies, "Ignore" / "None"
    leave the encodings alone (the parser has already dropped the weights for "None").
The tokenizer (a transformers T5TokenizerFast: its sentencepiece vocabulary is a data file) is handed in."""
import torch

from . import parsing

EOS_ID, PAD_ID = 1, 0
_EMPHASIS_MODES = ("Original", "No norm", "Ignore", "None")
_BRACKET_GAIN = {"(": 1.1, "]": 1.1, ")": 1 / 1.1, "[": 1 / 1.1}


class PromptChunk:
    """one row handed to the encoder: token ids and the per-token emphasis weights (same length)"""

    def __init__(self, tokens=None, multipliers=None):
        self.tokens = list(tokens or [])
        self.multipliers = list(multipliers or [])

    def padded(self, length):
        extra = max(0, length - len(self.tokens))
        return self.tokens + [PAD_ID] * extra, self.multipliers + [1.0] * extra


def _bracket_multiplier(text):
    """net emphasis of the brackets inside a vocabulary entry (t5_engine.py:38-53 keeps that table; nothing on the T5 path reads it, it is part of
    the object's surface)"""
    gain = 1.0
    for ch in text:
        gain *= _BRACKET_GAIN.get(ch, 1.0)
    return gain


class T5TextProcessingEngine:
    def __init__(self, text_encoder, tokenizer, emphasis_name="Original", min_length=256):
        if emphasis_name not in _EMPHASIS_MODES:
            raise ValueError(f"unknown emphasis mode {emphasis_name}")
        self.text_encoder = text_encoder.transformer
        self.tokenizer = tokenizer
        self.emphasis_name = emphasis_name
        self.min_length = min_length
        self.id_end, self.id_pad = EOS_ID, PAD_ID
        vocab = tokenizer.get_vocab()
        self.comma_token = vocab.get(",</w>")
        gains = ((ident, _bracket_multiplier(text)) for text, ident in vocab.items() if any(ch in text for ch in "()[]"))
        self.token_mults = {ident: gain for ident, gain in gains if gain != 1.0}

    # ---- tokens ---------------------------------------------------------------------------------------------------------------------------
    def tokenize(self, texts):
        return self.tokenizer(texts, truncation=False, add_special_tokens=False)["input_ids"]

    def tokenize_line(self, line):
        """-> (rows as PromptChunk, number of tokens incl. one EOS per row and excluding padding)"""
        pieces = parsing.parse_prompt_attention(line, self.emphasis_name)
        ids_per_piece = self.tokenize([text for text, _ in pieces])
        rows, open_row = [], PromptChunk()
        for ids, (text, weight) in zip(ids_per_piece, pieces):
            if text == "BREAK" and weight == -1:
                rows.append(open_row)
                open_row = PromptChunk()
            else:
                open_row.tokens.extend(ids)
                open_row.multipliers.extend([weight] * len(ids))
        if open_row.tokens or not rows:
            rows.append(open_row)
        counted = 0
        for row in rows:
            row.tokens.append(EOS_ID)
            row.multipliers.append(1.0)
            counted += len(row.tokens)
            row.tokens, row.multipliers = row.padded(self.min_length)
        return rows, counted

    # ---- encodings ------------------------------------------------------------------------------------------------------------------------
    def encode_with_transformers(self, tokens):
        return self.text_encoder(input_ids=tokens)

    def process_tokens(self, batch_tokens, batch_multipliers):
        z = self.encode_with_transformers(torch.asarray(batch_tokens))
        if self.emphasis_name not in ("Original", "No norm"):
            return z
        weights = torch.asarray(batch_multipliers).to(z).unsqueeze(-1)
        weighted = z * weights
        if self.emphasis_name == "Original":       # keep the encoding's mean where it was (emphasis.py:34-45)
            weighted = weighted * (z.mean() / weighted.mean())
        return weighted

    def _encode_line(self, line):
        rows, _ = self.tokenize_line(line)
        width = max(len(r.tokens) for r in rows)
        return [self.process_tokens([toks], [mults])[0] for toks, mults in (r.padded(width) for r in rows)]

    def __call__(self, texts):
        done = {}
        out = []
        for line in texts:
            if line not in done:
                done[line] = self._encode_line(line)
            out.extend(done[line])
        return torch.stack(out)
