"""Where a kernel's scratch traffic sits: tools/spill_map.py file.s kernel-name-substring -> run-length map of scratch stores/loads, barriers and MFMAs."""
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
i = [m for m in range(len(s)) if s.startswith(key, m) and s[m + len(key)] == ':' ][0]
j = s.index('.Lfunc_end', i)
k = s[i:j].split('\n')
print(len(k), 'lines')
ev = []
for n, l in enumerate(k):
    if 'scratch_store' in l: ev.append((n, 'ST'))
    elif 'scratch_load' in l: ev.append((n, 'LD'))
    elif 's_barrier' in l: ev.append((n, 'BAR'))
    elif 'v_mfma_f32_32x32' in l: ev.append((n, 'M32'))
    elif 'v_mfma_f32_16x16' in l: ev.append((n, 'M16'))
out = []; last = None; cnt = 0; start = 0
for n, t in ev:
    if t == last: cnt += 1
    else:
        if last: out.append(f"{last}x{cnt}@{start}")
        last = t; cnt = 1; start = n
out.append(f"{last}x{cnt}@{start}")
print(' '.join(out))
