"""Samples the GPU's shader clock and package power (sysfs, read-only) while a command runs, and prints their distribution.
usage: python tools/clock_probe.py -- python bench.py --steps 200 ...      (development aid: is the overlapped step clock / power limited?)"""
import glob, os, statistics, subprocess, sys, time

def read(path):
    try:
        return open(path).read()
    except OSError:
        return ""

def current_mhz(txt):
    for line in txt.splitlines():
        if "*" in line:
            for tok in line.replace("*", " ").split():
                if tok.lower().endswith("mhz"):
                    return float(tok[:-3])
    return None

cmd = sys.argv[sys.argv.index("--") + 1:]
cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
print("clock files:", cards, flush=True)
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) + sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
freq_in = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
print("power files:", hw, "freq files:", freq_in, flush=True)
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
clk, pw, fq = {c: [] for c in cards}, {h: [] for h in hw}, {f: [] for f in freq_in}
t0 = time.time()
while p.poll() is None:
    for c in cards:
        v = current_mhz(read(c))
        if v is not None:
            clk[c].append((time.time() - t0, v))
    for h in hw:
        t = read(h).strip()
        if t.isdigit():
            pw[h].append((time.time() - t0, int(t) / 1e6))
    for f in freq_in:
        t = read(f).strip()
        if t.isdigit():
            fq[f].append((time.time() - t0, int(t) / 1e6))
    time.sleep(0.02)
out = p.stdout.read()
def stats(name, series, tail):
    # the last `tail` seconds before the command ended = the timed region of a long bench run
    if not series:
        return
    tend = series[-1][0]
    v = [x for t, x in series if t >= tend - tail]
    if v:
        print(name, "n", len(v), "min", min(v), "median", statistics.median(v), "max", max(v), flush=True)
tail = float(os.environ.get("PROBE_TAIL_S", "3"))
for c, s in clk.items():
    stats("sclk MHz " + c, s, tail)
for f, s in fq.items():
    stats("freq1 MHz " + f, s, tail)
for h, s in pw.items():
    stats("power W " + h, s, tail)
print(out[-400:])
