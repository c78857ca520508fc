#!/bin/bash
# Socket power and shader clock (rocm-smi, 5 Hz) while (a) the sampling bench, (b) a pure MFMA stream, (c) the hot f16mx probe loop run: the direct
# reading behind DESIGN 6.0's "the GEMM family is bound by the chip's power management".   gpurun --timeout 900 -- 'bash tools/power_trace.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/power_trace.log
: > $L
poll() {   # $1 = tag; polls until the file /tmp/poll_stop exists
  while [ ! -e /tmp/poll_stop ]; do
    p=$(rocm-smi -P -c -M 2>/dev/null | grep -iE "power|sclk" | tr -s ' ' | tr '\n' ';')
    echo "$1 $(date +%s.%N | cut -c1-14) $p" >> $L
    sleep 0.2
  done
}
run() {  # tag, command
  rm -f /tmp/poll_stop; poll "$1" & PP=$!
  eval "$2" > /tmp/run_$1.log 2>&1
  touch /tmp/poll_stop; wait $PP
}
rocm-smi -P -c -M --showenergycounter > gpurun_out/power_idle.txt 2>&1
run idle "sleep 2"
run bench "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra"
run mfma "tools/native/mfma_mix_bench; tools/native/mfma_mix_bench"
run probe_mx "(cd tools/native && PROBE_WKBLK=1 ./kernel_probe mx 16 200)"
run train "python bench.py --mode train --steps 6 --warmup 1 --no-cpu-baseline --no-roofline"
python - <<'P'
import re, collections
acc = collections.defaultdict(list)
for l in open('gpurun_out/power_trace.log'):
    tag = l.split()[0]
    pw = re.search(r'(?:Socket|Average)[^;]*?Power \(W\): ([\d.]+)', l)
    ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', l) or re.search(r'sclk[^;]*?\((\d+)Mhz\)', l)
    cap = re.search(r'Max Graphics Package Power \(W\): ([\d.]+)', l)
    acc[tag].append((float(pw.group(1)) if pw else None, int(ck.group(1)) if ck else None, float(cap.group(1)) if cap else None))
for tag, v in acc.items():
    p = [a for a, _, _ in v if a is not None]; c = [b for _, b, _ in v if b is not None]; cap = [x for _, _, x in v if x is not None]
    if p:
        p2 = sorted(p)[len(p) // 4:]          # upper three quarters: drops the start-up samples
        print(f"{tag:9s}: {len(v):3d} samples | socket power W: mean of the upper 3/4 {sum(p2) / len(p2):7.1f}  max {max(p):7.1f} | sclk MHz: "
              f"{(sum(c) / len(c)) if c else float('nan'):7.0f} (min {min(c) if c else 0}, max {max(c) if c else 0}) | cap W {cap[0] if cap else 'n/a'}")
    else:
        print(tag, 'no power samples parsed; first line:', open('gpurun_out/power_trace.log').readline()[:300])
P
head -3 $L | cut -c1-400
