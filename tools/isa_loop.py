#!/usr/bin/env python
"""Print the instruction skeleton (mfma / ds_read / LDS-DMA / waits / barriers / scratch / branches) of the basic blocks of one kernel.
usage: python tools/isa_loop.py <file.hip> '<demangled-name-substring>' [min_mfma_per_block]"""
import os, re, subprocess, sys, tempfile
src, pat = sys.argv[1], sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = tempfile.mkdtemp()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-save-temps", "-c",
                os.path.abspath(src), "-I" + os.path.dirname(os.path.abspath(src)), "-o", os.path.join(d, "x.o")], cwd=d, capture_output=True)
sfile = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
s = open(os.path.join(d, sfile)).read()
syms = re.findall(r"^(_Z\w+):", s, re.M)
dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
for sym, dn in zip(syms, dem):
    if pat in dn:
        break
else:
    raise SystemExit("no kernel matches")
print("#", dn[:160])
i = s.index(sym + ":"); j = s.index(".Lfunc_end", i)
blk, out = None, {}
for ln in s[i:j].split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m:
        blk = m.group(1); out[blk] = []; continue
    if blk is None: continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    if op.startswith("v_mfma"): out[blk].append("M")
    elif op.startswith("ds_read"): out[blk].append("r")
    elif op.startswith("ds_write"): out[blk].append("w")
    elif op.startswith("buffer_load") and " lds" in t: out[blk].append("D")
    elif op.startswith(("buffer_load", "global_load")): out[blk].append("L")
    elif op.startswith(("buffer_store", "global_store")): out[blk].append("S")
    elif op.startswith("scratch_"): out[blk].append("!scratch!")
    elif op == "s_waitcnt": out[blk].append("[" + t[len("s_waitcnt"):].strip() + "]")
    elif op == "s_barrier": out[blk].append("|BAR|")
    elif op.startswith(("s_cbranch", "s_branch")): out[blk].append("<" + t + ">")
    elif op.startswith("v_"): out[blk].append("v")
    elif op.startswith("s_"): out[blk].append("s")
for b, seq in out.items():
    if seq.count("M") >= minm:
        # compress runs
        comp, prev, n = [], None, 0
        for x in seq + [None]:
            if x == prev: n += 1
            else:
                if prev is not None: comp.append(prev if n == 1 else f"{prev}{n}")
                prev, n = x, 1
        print(b, " ".join(comp))
