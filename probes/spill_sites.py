"""Where does a kernel spill?  Lists the scratch_load / scratch_store instructions of every kernel in a hipcc -S listing whose
mangled name contains the given pattern, and whether each sits inside a loop (between a label and a backward branch to it).
    python probes/kernel_meta.py <file.hip>            (writes /tmp/kmeta_<file>.s)
    python probes/spill_sites.py /tmp/kmeta_<file>.s <pattern>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
for a in starts:
    name = lines[a].split(":")[0]
    if pat not in name:
        continue
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[a:b]
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k))
    print(name, "instructions", len(body), "loops", loops, "mfma", sum("v_mfma" in l for l in body))
    for k, l in enumerate(body):
        if "scratch_" in l:
            print("   ", k, l.strip()[:80], "<-- IN LOOP" if any(x <= k <= y for x, y in loops) else "")
