"""Dev tool: scans one kernel of a hipcc -S listing for stall patterns hipcc tends to produce:
  * an LDS / global read whose wait follows within a few instructions (a serialised round trip),
  * exec-masked regions (s_and_saveexec) -- each one a pair of scalar-branch sequences around a few vector instructions,
and prints the instruction mix.  usage: isa_audit.py listing.s kernel-name-substring [first-line last-line]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = [l.strip() for l in lines[start:end]]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
ins = [(i, l) for i, l in enumerate(body) if lo <= i < hi and l and not l.startswith((';', '.')) and not l.endswith(':')]
mix = {}
for _, l in ins:
  op = l.split()[0]
  key = ('mfma' if 'mfma' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write') else
         'vmem' if op.startswith(('buffer_', 'global_', 'scratch_')) else 'waitcnt' if op == 's_waitcnt' else
         'branch' if op.startswith('s_cbranch') else 'salu' if op.startswith('s_') else
         'valu_quarter' if op in ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_mul_lo_u32', 'v_mul_hi_u32') else 'valu' if op.startswith('v_') else 'other')
  mix[key] = mix.get(key, 0) + 1
print('instructions', len(ins), mix)
ser = []
for n, (i, l) in enumerate(ins):
  if l.startswith(('ds_read', 'global_load', 'buffer_load')):
    for j in range(n + 1, min(n + 4, len(ins))):
      w = ins[j][1]
      if w.startswith('s_waitcnt') and (('lgkmcnt(0)' in w and l.startswith('ds_')) or ('vmcnt(0)' in w and not l.startswith('ds_'))):
        ser.append((i, l.split()[0]))
        break
print('reads waited for within 3 instructions:', len(ser), [f'{i}:{op}' for i, op in ser[:60]])
print('exec-masked regions:', sum(1 for _, l in ins if l.startswith('s_and_saveexec')))
