"""Prompt-attention syntax -- mirror of backend/text_processing/parsing.py (`parse_prompt_attention` :22-85): `(text)` x1.1, `[text]` /1.1,
`(text:1.3)` explicit weight, backslash escapes, the BREAK keyword.  Host-side text processing; returns [[text, weight], ...] with ["BREAK", -1]
separators, adjacent runs of equal weight merged."""
import re

_PIECE = re.compile(r"\\[()\[\]\\]|\\|\(|\[|:\s*([+-]?[.\d]+)\s*\)|\)|\]|[^\\()\[\]:]+|:")
_BREAK = re.compile(r"\s*\bBREAK\b\s*", re.S)
ROUND, SQUARE = 1.1, 1 / 1.1


def parse_prompt_attention(text, emphasis="Original"):
    if emphasis == "None":  # the mechanism is off: the whole text is literal
        return [[text, 1.0]]
    runs = []                      # [text, weight]
    open_round, open_square = [], []   # indices into `runs` where an unclosed bracket began

    def scale_from(start, factor):
        for r in runs[start:]:
            r[1] *= factor
    for m in _PIECE.finditer(text):
        piece, explicit = m.group(0), m.group(1)
        if piece[0] == "\\":
            runs.append([piece[1:], 1.0])          # escaped bracket / backslash: literal (a lone backslash contributes nothing)
        elif piece == "(":
            open_round.append(len(runs))
        elif piece == "[":
            open_square.append(len(runs))
        elif explicit is not None and open_round:
            scale_from(open_round.pop(), float(explicit))
        elif piece == ")" and open_round:
            scale_from(open_round.pop(), ROUND)
        elif piece == "]" and open_square:
            scale_from(open_square.pop(), SQUARE)
        else:
            for i, part in enumerate(_BREAK.split(piece)):
                if i:
                    runs.append(["BREAK", -1])
                runs.append([part, 1.0])
    for start in open_round:       # brackets never closed apply to everything after them
        scale_from(start, ROUND)
    for start in open_square:
        scale_from(start, SQUARE)
    if not runs:
        return [["", 1.0]]
    merged = [runs[0]]
    for t, w in runs[1:]:
        if w == merged[-1][1]:
            merged[-1][0] += t
        else:
            merged.append([t, w])
    return merged
