"""Renderer of the viewer HOOK (second half): one position as an SVG string.

``Game.graphic`` (reference game.py:47-75) and ``BatchedSelfPlay.watch`` push ``board_svg(...)`` plus a status line to anything with
``update_board(svg, status_text)``. The reference renders with ``cchess.svg``, which is absent here: :func:`board_svg` draws the
position itself (grid, river, palace, piece discs with the FEN letters). The HTTP window that can sit behind the hook is NOT part
of the package (out of scope): ``examples/viewer.py``. Nothing on the hot path imports this module.
"""
from __future__ import annotations

_LETTER = {1: "P", 2: "C", 3: "R", 4: "N", 5: "B", 6: "A", 7: "K"}


def board_svg(squares, last_move=None, cell: int = 64) -> str:
    """SVG of a position: ``squares`` = 90 piece codes (0 empty, red = type, black = type + 8), red at the bottom.
    ``last_move`` = (from_square, to_square) highlights the move just played."""
    w, h, m = 8 * cell, 9 * cell, cell
    xy = lambda s: (m + (s % 9) * cell, m + (9 - s // 9) * cell)
    out = [f'<svg xmlns="http://www.w3.org/2000/svg" viewBox="0 0 {w + 2 * m} {h + 2 * m}" width="{w + 2 * m}" height="{h + 2 * m}">',
           f'<rect width="100%" height="100%" fill="#f0d9a8"/>']
    for r in range(10):
        out.append(f'<line x1="{m}" y1="{m + r * cell}" x2="{m + w}" y2="{m + r * cell}" stroke="#333"/>')
    for f in range(9):
        if f in (0, 8):
            out.append(f'<line x1="{m + f * cell}" y1="{m}" x2="{m + f * cell}" y2="{m + h}" stroke="#333"/>')
        else:  # the river interrupts the inner files
            out.append(f'<line x1="{m + f * cell}" y1="{m}" x2="{m + f * cell}" y2="{m + 4 * cell}" stroke="#333"/>')
            out.append(f'<line x1="{m + f * cell}" y1="{m + 5 * cell}" x2="{m + f * cell}" y2="{m + h}" stroke="#333"/>')
    for top in (0, 7):  # palaces
        x0, x1, y0, y1 = m + 3 * cell, m + 5 * cell, m + top * cell, m + (top + 2) * cell
        out.append(f'<line x1="{x0}" y1="{y0}" x2="{x1}" y2="{y1}" stroke="#333"/><line x1="{x1}" y1="{y0}" x2="{x0}" y2="{y1}" stroke="#333"/>')
    if last_move is not None:
        for s in last_move:
            x, y = xy(int(s))
            out.append(f'<rect x="{x - cell // 2}" y="{y - cell // 2}" width="{cell}" height="{cell}" fill="#7fc97f" fill-opacity="0.45"/>')
    for s in range(90):
        pc = int(squares[s])
        if pc:
            x, y = xy(s)
            red = pc < 8
            out.append(f'<circle cx="{x}" cy="{y}" r="{cell * 0.42:.0f}" fill="#fff8e7" stroke="{"#c00" if red else "#111"}" stroke-width="3"/>'
                       f'<text x="{x}" y="{y + cell * 0.14:.0f}" font-size="{cell * 0.42:.0f}" text-anchor="middle" fill="{"#c00" if red else "#111"}" '
                       f'font-family="sans-serif">{_LETTER[pc & 7] if red else _LETTER[pc & 7].lower()}</text>')
    out.append("</svg>")
    return "".join(out)
