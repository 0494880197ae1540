#!/usr/bin/env python3
"""Count instructions per kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import collections
import re
import sys


def main(path, pattern=""):
    lines = open(path).read().split("\n")
    name, cnt = None, None
    out = []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cnt = m.group(1), collections.Counter()
            out.append((name, cnt))
            continue
        if name is None:
            continue
        if ln.startswith(".Lfunc_end"):
            name = None
            continue
        t = ln.strip()
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        cnt[t.split()[0]] += 1
    for name, cnt in out:
        if pattern not in name:
            continue
        g = collections.Counter()
        for k, v in cnt.items():
            key = ("v_pk" if k.startswith("v_pk") else "trans" if k in ("v_log_f32", "v_exp_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32")
                   else "valu" if k.startswith("v_") else "ds" if k.startswith("ds_") else
                   "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if k.startswith("s_") else "other")
            g[key] += v
        print(name[:90])
        print("  total", sum(cnt.values()), dict(g))
        print("  top:", ", ".join("%s=%d" % kv for kv in cnt.most_common(28)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
