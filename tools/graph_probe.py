#!/usr/bin/env python3
"""Why is hipGraph replay of the training step slower than eager launching?

  python tools/graph_probe.py eager3|eager1|graph3|graph1 [steps=20]

Times the configs[1] step four ways, one per process — eager / graph replay, each with the three-stream schedule and with
everything on one stream (GLOWTTS_SIDE_STREAM=0) — and dumps the captured graphs (hipGraphDebugDotPrint): nodes, edges, how
many nodes have more than one predecessor / successor (the cross-stream joins), the length of the longest dependency chain.
If replay(3 streams) ~ replay(1 stream) ~ eager(1 stream), the graph executor is running the captured branches one after
another: the step's 18 % gain from overlapping streams is what replay loses."""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def dot_stats(path):
    txt = open(path).read()
    edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
    nodes = set(re.findall(r'^\s*"?([\w.]+)"?\s*\[', txt, flags=re.M))
    for a, b in edges:
        nodes.add(a)
        nodes.add(b)
    preds, succs = {}, {}
    for a, b in edges:
        preds.setdefault(b, []).append(a)
        succs.setdefault(a, []).append(b)
    depth = {}
    order = list(nodes)
    # longest chain by repeated relaxation over a topological order (Kahn)
    indeg = {n: len(preds.get(n, [])) for n in nodes}
    ready = [n for n in nodes if indeg[n] == 0]
    while ready:
        n = ready.pop()
        depth[n] = 1 + max((depth[q] for q in preds.get(n, [])), default=0)
        for m in succs.get(n, []):
            indeg[m] -= 1
            if indeg[m] == 0:
                ready.append(m)
    kinds = {}
    for label in re.findall(r'label="([^"]*)"', txt):
        k = label.split("\\n")[0].split("(")[0].strip()[:40]
        kinds[k] = kinds.get(k, 0) + 1
    return {"nodes": len(nodes), "edges": len(edges), "joins(>1 pred)": sum(len(v) > 1 for v in preds.values()),
            "forks(>1 succ)": sum(len(v) > 1 for v in succs.values()), "longest_chain": max(depth.values(), default=0),
            "roots": sum(1 for n in nodes if not preds.get(n)), "top_labels": sorted(kinds.items(), key=lambda kv: -kv[1])[:6]}


def main():
    """One measurement per process (an eager step AFTER a capture faulted once inside torch's embedding backward: keep
    captures last): mode = eager3 | eager1 | graph3 | graph1."""
    mode = sys.argv[1] if len(sys.argv) > 1 else "graph3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sys.argv = [sys.argv[0]]                        # bench.parse() reads the defaults: BASELINE configs[1]
    args = bench.parse()
    os.environ["GLOWTTS_SIDE_STREAM"] = "1" if mode.endswith("3") else "0"
    dev = torch.device("cuda", 0)
    from glow_tts_train import _hip
    from glow_tts_train.train import GraphedTrainStep, train_batch

    _hip.load()
    model, opt, batch, cfg = bench.build_workload(args, dev, 0)

    def timed(fn, n=steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    if mode.startswith("eager"):
        print(f"{mode}: {timed(lambda: train_batch(model, opt, batch, cfg.grad_clip, None)):6.2f} ms/step", flush=True)
        return
    g = GraphedTrainStep(model, opt, cfg.grad_clip, batch, warmup=2)
    print(f"{mode}: {timed(lambda: g()):6.2f} ms/step (hipGraph replay)", flush=True)
    try:                                            # structure of the captured graph, when the runtime can print it
        g2 = torch.cuda.CUDAGraph()
        g2.enable_debug_mode()
        with torch.cuda.graph(g2):
            train_batch(model, opt, g.static, cfg.grad_clip)
        dot = os.path.join("/tmp", f"graph_{mode}.dot")
        g2.debug_dump(dot)
        if os.path.exists(dot):
            print("   graph:", dot_stats(dot), flush=True)
        else:
            print("   hipGraphDebugDotPrint wrote no file")
    except Exception as exc:
        print("   graph dump failed:", type(exc).__name__, exc)
    sys.stdout.flush()
    os._exit(0)                                     # no teardown of captured graphs


if __name__ == "__main__":
    main()
