#!/usr/bin/env python3
"""No packed fp32 instruction of the shipped library may route a HIGH source half into its LOW lane (`op_sel:[..1..]`): beside another
wave's MFMAs such an instruction returns wrong low halves on MI355X (tools/pk_f32_hazard.hip).  The library is built
with -fno-slp-vectorize, which is what formed them; this check disassembles every object of relightableavatar_amd/csrc and lists the
packed fp32 instructions by form.  Exit status 1 if an unsafe form is present.   python3 tools/check_packed_fp32.py [obj.o ...]"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def packed_fp32_forms(obj):
    """Counter of (mnemonic, 'op_sel' | 'op_sel_hi only' | 'plain') over the device code of one object (empty if it has none)"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'dev.co')
        if subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', obj], capture_output=True).returncode or not os.path.exists(fat):
            return Counter()
        if subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                           f'--output={co}'], capture_output=True).returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            return Counter()
        asm = subprocess.run([f'{LLVM}/llvm-objdump', '-d', co], capture_output=True, text=True).stdout
    c = Counter()
    for l in asm.split('\n'):
        m = re.search(r'\b(v_pk_(?:mul|add|fma)_f32)\b(.*?)(?://|$)', l)
        if not m:
            continue
        ops = m.group(2)
        low = re.search(r'op_sel:\[([01,]+)\]', ops)            # op_sel (without _hi): source halves of the LOW lane
        if low and '1' in low.group(1):
            c[(m.group(1), 'op_sel')] += 1
        elif 'op_sel_hi' in ops:
            c[(m.group(1), 'op_sel_hi only')] += 1
        else:
            c[(m.group(1), 'plain')] += 1
    return c


def check(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, 'relightableavatar_amd', 'csrc', '*.o')))
    report, unsafe = {}, 0
    for o in objs:
        c = packed_fp32_forms(o)
        if c:
            report[os.path.basename(o)] = dict((f'{k[0]} {k[1]}', v) for k, v in sorted(c.items()))
            unsafe += sum(v for k, v in c.items() if k[1] == 'op_sel')
    return unsafe, report


if __name__ == '__main__':
    unsafe, report = check(sys.argv[1:] or None)
    for o, r in report.items():
        print(o, r)
    print('unsafe packed fp32 instructions (a high source half in the low lane):', unsafe)
    sys.exit(1 if unsafe else 0)
