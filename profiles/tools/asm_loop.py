"""Print the MFMA main loop of one kernel from a hipcc -S listing, with runs of MFMAs collapsed.

  python profiles/tools/asm_loop.py file.s <substring of the kernel symbol> [max lines]
"""
import sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(';')[0].strip().endswith(':')][0]
end = start
while not lines[end].startswith('.Lfunc_end'): end += 1
body = lines[start:end]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
# the loop: the innermost loop header before the first MFMA .. the backward branch after the last MFMA
j = mf[0]
while 'Loop Header' not in body[j]: j -= 1
k = mf[-1]
while 's_cbranch' not in body[k] and 's_branch' not in body[k]: k += 1
out, m = [], 0
for l in body[j:k + 1]:
    t = l.strip()
    if not t or t.startswith(';'): continue
    if 'v_mfma' in t:
        m += 1
        continue
    if m:
        out.append(f"        [mfma x{m}]")
        m = 0
    out.append(l.split(';')[0].rstrip()[:100])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
print(f"{len(mf)} MFMAs, {k - j} lines in the loop")
print('\n'.join(out[:n]))
