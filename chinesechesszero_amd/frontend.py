"""Board viewer hook (second half): a small HTTP window that shows ONE selected board.

Mirror of the surface of reference frontend.py that its game loop uses (game.py:47-75): ``get_chess_window()`` returns a
window object with ``update_board(svg, status_text)``; the window serves ``/`` (a page that polls), ``/board`` (JSON with the
keys ``svg``, ``status``, ``timestamp``, frontend.py:120-136) and ``/events`` (server-sent events). The reference renders with
``cchess.svg``, which is absent here: :func:`board_svg` draws the position itself (grid, river, palace, piece discs with the
FEN letters). Stdlib only, off unless asked for (``--show`` in the reference); nothing on the hot path imports it.
"""
from __future__ import annotations

import json
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

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


_PAGE = """<!doctype html><meta charset="utf-8"><title>cczero-mi355x board</title>
<body style="font-family:sans-serif;text-align:center"><div id="status"></div><div id="board"></div>
<script>async function tick(){try{const r=await fetch('/board');const d=await r.json();
document.getElementById('status').textContent=d.status;document.getElementById('board').innerHTML=d.svg;}catch(e){}
setTimeout(tick,500);}tick();</script></body>"""


class ChessWindow:
    """``update_board(svg_content, status_text)`` as reference frontend.py:328-355; ``start()`` / ``stop()`` as :309-362."""

    def __init__(self, host: str = "127.0.0.1", port: int = 8000):
        self.host, self.port = host, port
        self.server = None
        self.thread = None
        self._lock = threading.Lock()
        self._state = {"svg": "", "status": "", "timestamp": 0.0}
        self.updates = 0

    def start(self):
        window = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, *a):  # quiet, like the reference's handler (frontend.py:209-215)
                pass

            def _send(self, body: bytes, ctype: str):
                self.send_response(200)
                self.send_header("Content-Type", ctype)
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def do_GET(self):
                if self.path.startswith("/board"):
                    with window._lock:
                        body = json.dumps(window._state).encode("utf-8")
                    self._send(body, "application/json; charset=utf-8")
                elif self.path.startswith("/events"):
                    self.send_response(200)
                    self.send_header("Content-Type", "text/event-stream")
                    self.send_header("Cache-Control", "no-cache")
                    self.end_headers()
                    seen = -1.0
                    try:
                        for _ in range(1200):  # bounded: a viewer reconnects
                            with window._lock:
                                st = dict(window._state)
                            if st["timestamp"] != seen:
                                seen = st["timestamp"]
                                self.wfile.write(b"data: " + json.dumps(st).encode("utf-8") + b"\n\n")
                                self.wfile.flush()
                            time.sleep(0.25)
                    except (BrokenPipeError, ConnectionResetError):
                        pass
                else:
                    self._send(_PAGE.encode("utf-8"), "text/html; charset=utf-8")

        self.server = ThreadingHTTPServer((self.host, self.port), Handler)
        self.port = self.server.server_address[1]  # port 0 = pick a free one
        self.thread = threading.Thread(target=self.server.serve_forever, daemon=True)
        self.thread.start()
        return self

    def update_board(self, svg_content, status_text_: str = ""):
        if hasattr(svg_content, "_repr_svg_"):
            svg_content = svg_content._repr_svg_()
        svg = str(svg_content)
        if not svg.lstrip().startswith("<svg"):  # plain text (e.g. str(board)): show it preformatted
            svg = "<pre>" + svg.replace("&", "&amp;").replace("<", "&lt;") + "</pre>"
        with self._lock:
            self._state = {"svg": svg, "status": status_text_, "timestamp": time.time()}
            self.updates += 1

    def stop(self):
        if self.server is not None:
            self.server.shutdown()
            self.server.server_close()
            self.server = None


_window = None


def get_chess_window(host: str = "127.0.0.1", port: int = 8000):
    """Singleton window (reference frontend.py:365-388); the browser is not opened automatically here."""
    global _window
    if _window is None:
        _window = ChessWindow(host, port).start()
    return _window
