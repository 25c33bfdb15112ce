#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: python tools/show_bench.py file.log"""
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
d = json.loads(line)
print(f"value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.3f}  kernel_ms {d['roofline']['kernel_ms']:.3f}  frac {d['roofline']['frac']:.4f}")
print(f"iteration {d['socm_ms_per_iter']:.3f} ms [{d.get('socm_iteration_mode')}]  eager {d.get('socm_ms_per_iter_eager')}  graph {d.get('socm_ms_per_iter_graph')}")
if "roofline_full_chip" in d:
    print(f"burst {d['roofline_full_chip']['kernel_ms']:.2f} ms frac {d['roofline_full_chip']['frac']:.3f}")
for s in d.get("secondary", []):
    print(f"  {s['workload'][:44]:44s} rollout {s['rollout_ms']:.3f} ms  iter {s['socm_ms_per_iter']:.3f} [{s.get('iteration_mode')}] eager {s.get('socm_ms_per_iter_eager'):.3f} graph {s.get('socm_ms_per_iter_graph')}")
if "cpu_baseline" in d:
    print(f"cpu {d['cpu_baseline']['value']:.4g}  x{d.get('speedup_vs_cpu_baseline'):.1f}")
