"""instruction mix per MFMA of one kernel in a gfx950 assembly listing (hipcc -S / -save-temps): python tools/isa_mix.py file.s kernel_name_substring"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
st = [i for i, l in enumerate(s) if l.startswith('_Z') and key in l and ': ;' in l][0]
en = [i for i, l in enumerate(s) if i > st and '.Lfunc_end' in l][0]
L = [l.strip() for l in s[st:en] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = collections.Counter(l.split()[0] for l in L)
nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
print(s[st][:100], 'mfma', nm, 'instructions per mfma %.2f' % (sum(c.values()) / nm))
valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
trans = sum(v for k, v in c.items() if k.split('_')[1] in ('exp', 'log', 'sin', 'cos', 'rcp', 'rsq', 'sqrt'))
print('VALU per mfma %.2f (transcendental %.2f)' % (valu / nm, trans / nm))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 18):
    print(f'   {k:32s}{v:7d} {v / nm:.3f}')
txt = '\n'.join(s)
m = re.search(r'\.name:\s+' + re.escape(s[st].split(':')[0]) + r'\n(?:.*\n){0,12}', txt)
for fld in ('vgpr_count', 'agpr_count', 'vgpr_spill_count', 'private_segment_fixed_size'):
    blk = txt[txt.find('.name:           ' + s[st].split(':')[0]) - 600: txt.find('.name:           ' + s[st].split(':')[0]) + 600]
    mm = re.search(fld + r':\s+(\d+)', blk)
    if mm: print('  ', fld, mm.group(1))
