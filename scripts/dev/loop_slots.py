#!/usr/bin/env python
"""Issue-slot model of every loop of a kernel (scripts/isa_cost.py's per-instruction classes, profiles/r02/valu_calibration.md):
a SIMD issues per 4-cycle slot either one 'slow' VALU op (SGPR source, compare / select, 64-bit integer, DPP, packed, a read-modify-write FMA
whose multiplicands share a register parity, transcendental = 2 slots) possibly paired with a plain op of another wave, or two plain ops of
two waves.  slots >= max(n_slow, n_valu / 2): a loop with n_slow > n_valu / 2 is bound by its slow ops.

    python scripts/dev/loop_slots.py <file.s> <kernel-name-substring> [min-instructions]"""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import isa_cost as IC

path, kern = sys.argv[1], sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^\w*" + re.escape(kern) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    t = labels.get(m.group(1) or m.group(2)) if m else None
    if t is None or t >= i:
        continue
    seg = [x.strip() for x in body[t:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
    if len(seg) < min_len:
        continue
    cat = collections.Counter()
    for x in seg:
        if x.startswith("v_"):
            cat[IC.cost(x)[1]] += 1
    n = sum(cat.values())
    slow = cat["sgpr source"] + cat["half-rate op"] + cat["fma operand-pair parity"] + cat["dpp"] + 2 * cat["trans"]
    print(f"loop {t}..{i}: {len(seg)} instr, {n} VALU, slow {slow} (sgpr {cat['sgpr source']}, half-rate {cat['half-rate op']}, parity {cat['fma operand-pair parity']}, trans {cat['trans']}), "
          f"slots >= max({slow}, {n / 2:.0f}) = {max(slow, n / 2):.0f}")
