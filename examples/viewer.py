"""EXAMPLE, not part of the package: a small HTTP window behind the viewer hook, showing ONE selected board.

The reference's browser viewer (frontend.py) is out of scope for the rollout path; what the package keeps is
the HOOK -- ``Game(viewer=...)``, ``BatchedSelfPlay.watch(board, viewer)``, ``CollectPipeline.run(is_shown=True, viewer=...)`` push
``(svg, status_text)`` to anything with ``update_board`` (reference game.py:47-75, frontend.py:328-355). This file is one such
thing: ``get_chess_window()`` returns a window that serves ``/`` (a page that polls), ``/board`` (JSON with the keys ``svg``,
``status``, ``timestamp``, frontend.py:120-136) and ``/events`` (server-sent events). Stdlib only.

    python -m chinesechesszero_amd.collect --boards 4096 --show      # picks this window up when run from the repository root
"""
from __future__ import annotations

import json
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

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
