#!/usr/bin/env python
"""Stub of the LLaVA + BERTScore reward server the reference talks to (callbacks.py:465-537): accepts the pickled request
{"images": [jpeg bytes], "queries": [[str]], "answers": [[str]]} on POST / and replies with a pickled dict
{"recall", "precision", "f1", "outputs"} of per-image values.  Scores are a deterministic function of the JPEG payload
(size-based) so tests are reproducible; `llava_vqa` requests ({"images", "queries"}) get {"outputs"} echoing "yes"."""
import pickle
import sys
from http.server import BaseHTTPRequestHandler, HTTPServer


class Handler(BaseHTTPRequestHandler):
    def do_POST(self):
        data = pickle.loads(self.rfile.read(int(self.headers["Content-Length"])))
        n = len(data["images"])
        if "answers" in data:
            rec = [[(len(b) % 1000) / 1000.0] for b in data["images"]]
            reply = {"recall": rec, "precision": [[r[0] / 2] for r in rec], "f1": [[r[0] / 3] for r in rec],
                     "outputs": [["a stub description"] for _ in range(n)]}
        else:
            reply = {"outputs": [["yes"] * len(q) for q in data["queries"]]}
        body = pickle.dumps(reply)
        self.send_response(200)
        self.send_header("Content-Length", str(len(body)))
        self.end_headers()
        self.wfile.write(body)

    def log_message(self, *a):
        pass


def serve(port=8085):
    srv = HTTPServer(("127.0.0.1", port), Handler)
    return srv


if __name__ == "__main__":
    serve(int(sys.argv[1]) if len(sys.argv) > 1 else 8085).serve_forever()
