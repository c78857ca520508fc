#!/usr/bin/env python
"""Which kernels of a translation unit changed between two builds?  Compares the instruction streams function by function
(labels normalised, comments and directives dropped) of two device assembly files, e.g.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -S --cuda-device-only gemm_bf16.hip -o before.s
    ... edit: add a new template instantiation / constexpr branch ...
    hipcc ... -o after.s ;  python tools/isa_diff.py before.s after.s

Used to add experimental kernel variants as NEW instantiations while proving that every validated kernel is unchanged
instruction for instruction (no GPU needed)."""
import re
import sys


def funcs(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if line.strip().startswith(".end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = re.sub(r";.*", "", line).strip()
        if t and not t.startswith("."):
            out[cur].append(re.sub(r"\.LBB\d+_", ".LBB_", t))
        elif t.startswith(".LBB"):
            out[cur].append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out


def main(a_path, b_path):
    a, b = funcs(a_path), funcs(b_path)
    changed = [k for k in a if k in b and a[k] != b[k]]
    for k in changed:
        print(f"CHANGED  {k[:110]}  ({len(a[k])} -> {len(b[k])} instructions)")
    for k in b:
        if k not in a:
            print(f"NEW      {k[:110]}  ({len(b[k])} instructions)")
    for k in a:
        if k not in b:
            print(f"REMOVED  {k[:110]}")
    print(f"{len(a)} -> {len(b)} kernels, {len(changed)} changed")
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
