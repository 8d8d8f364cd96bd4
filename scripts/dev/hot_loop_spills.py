"""Does a kernel's pixel loop carry spill traffic?  Lists, for every loop of the kernel (a backward branch to a label) that holds a v_rsq_f32
(the Huber weight: only the dense pixel loops have one), its instruction count and the v_readlane / v_writelane / scratch_ instructions in it.
usage: python scripts/dev/hot_loop_spills.py <file.s> <kernel-substring>"""
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t < i:
            loops.append((t, i))
for t, i in sorted(loops, key=lambda x: x[1] - x[0]):
    seg = [l.strip() for l in body[t:i + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    if not any("v_rsq_f32" in l for l in seg):
        continue
    valu = sum(1 for l in seg if l.startswith("v_"))
    spill = [l.split()[0] for l in seg if l.startswith(("v_readlane", "v_writelane", "scratch_", "buffer_load_dword", "buffer_store_dword"))]
    print(f"loop {t}..{i}: {len(seg)} instructions, {valu} VALU, spill-like: {len(spill)} {sorted(set(spill))}")
