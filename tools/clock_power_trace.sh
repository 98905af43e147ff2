#!/bin/bash
# Clock / power trace of the headline bench (profiles/rNN_clock_power.txt): rocm-smi sampled every 250 ms while
# `bench.py --steps 120` runs, then a summary of the sclk and socket power the chip HOLDS during the timed steps.
# The dense bf16 MFMA peak of 2.5 PFLOP/s is 256 CU x 4 SIMD x 1024 FLOP/clk at 2.4 GHz; what the step can reach scales
# with the clock the chip sustains under this load.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/clock_power_raw.txt
: > $OUT
( while true; do echo "t=$(date +%s.%N)" >> $OUT; rocm-smi -P -c -u --json >> $OUT 2>/dev/null; echo >> $OUT; sleep 0.25; done ) &
SMI=$!
sleep 2
python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r03/clock_power_bench.json 2> gpurun_out/r03/clock_power_bench.err
sleep 1
kill $SMI
python3 - <<'PY'
import json, re
raw = open("gpurun_out/r03/clock_power_raw.txt").read().split("t=")[1:]
rows = []
for blk in raw:
    lines = blk.strip().split("\n")
    try:
        t = float(lines[0]); d = json.loads(lines[1])
    except Exception:
        continue
    c = d.get("card0", {})
    def num(pat):
        for k, v in c.items():
            if re.search(pat, k, re.I):
                m = re.search(r"[-+]?\d+\.?\d*", str(v))
                if m: return float(m.group())
        return None
    rows.append((t, num(r"sclk clock speed|sclk"), num(r"power"), num(r"GPU use")))
t0 = rows[0][0]
b = json.load(open("gpurun_out/r03/clock_power_bench.json"))
busy = [r for r in rows if r[3] is not None and r[3] >= 90]
print(f"# rocm-smi trace while `bench.py --steps 120 --warmup 10` ran: {b['ms_per_step']} ms/step, {b['value']} tokens/s, mfma_utilisation_step {b['mfma_utilisation_step']}")
print(f"# {len(rows)} samples, {len(busy)} with GPU use >= 90 %")
if busy:
    sc = sorted(r[1] for r in busy if r[1]); pw = sorted(r[2] for r in busy if r[2])
    med = lambda v: v[len(v) // 2]
    print(f"# while busy: sclk median {med(sc):.0f} MHz (min {sc[0]:.0f}, max {sc[-1]:.0f}); socket power median {med(pw):.0f} W (max {pw[-1]:.0f})")
    print(f"# dense bf16 MFMA peak at the median busy clock: {2.5e3 * med(sc) / 2400:.0f} TFLOP/s (2500 at 2400 MHz)")
print("# t(s)  sclk(MHz)  power(W)  use(%)")
for t, s, p, u in rows:
    print(f"{t - t0:7.2f}  {s}  {p}  {u}")
PY
