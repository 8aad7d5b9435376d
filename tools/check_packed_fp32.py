#!/usr/bin/env python3
"""No packed fp32 instruction of the shipped library may route a HIGH source half into its LOW lane (`op_sel:[..1..]`): beside another
wave's MFMAs such an instruction returns wrong low halves on MI355X (tools/pk_f32_hazard.hip).  The library is built
with -fno-slp-vectorize, which is what formed them; this check disassembles every object of relightableavatar_amd/csrc and lists the
packed fp32 instructions by form.  Exit status 1 if an unsafe form is present — or if the check could not be made: a missing / failing
llvm tool, or an object built from a .hip / device .cpp source without extractable device code for the architecture, is an ERROR, not a
pass (the gate must not fail open).   python3 tools/check_packed_fp32.py [--arch gfx950] [obj.o ...]"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CheckError(RuntimeError):
    """the check could not be made"""


def _run(cmd, **kw):
    try:
        return subprocess.run(cmd, capture_output=True, **kw)
    except OSError as ex:           # the tool is not there
        raise CheckError(f'{cmd[0]}: {ex}')


def has_fatbin(obj):
    r = _run([f'{LLVM}/llvm-objdump', '-h', obj], text=True)
    if r.returncode:
        raise CheckError(f'llvm-objdump -h {obj}: {r.stderr.strip()[-300:]}')
    return '.hip_fatbin' in r.stdout


def packed_fp32_forms(obj, arch='gfx950'):
    """Counter of (mnemonic, 'op_sel' | 'op_sel_hi only' | 'plain') over the device code of one object; None for a host-only object
    (no .hip_fatbin section).  Raises CheckError when a tool fails or the fat binary holds no code object for `arch`."""
    if not has_fatbin(obj):
        return None
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'dev.co')
        r = _run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', obj])
        if r.returncode or not os.path.exists(fat):
            raise CheckError(f'llvm-objcopy could not extract .hip_fatbin of {obj}: {r.stderr.decode()[-300:]}')
        r = _run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', f'--targets=hipv4-amdgcn-amd-amdhsa--{arch}', f'--output={co}'])
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            raise CheckError(f'{obj}: no device code object for {arch} in its fat binary (clang-offload-bundler: {r.stderr.decode()[-300:]})')
        r = _run([f'{LLVM}/llvm-objdump', '-d', co], text=True)
        if r.returncode or 's_endpgm' not in r.stdout:
            raise CheckError(f'{obj}: llvm-objdump produced no {arch} disassembly')
        asm = r.stdout
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


def check(objs=None, arch='gfx950'):
    """(unsafe count, {object: forms}) — raises CheckError if any object could not be checked, or if an object whose source is a .hip
    file (always device code) turns out host-only"""
    objs = objs or sorted(glob.glob(os.path.join(ROOT, 'relightableavatar_amd', 'csrc', '*.o')))
    if not objs:
        raise CheckError('no objects to check')
    report, unsafe = {}, 0
    for o in objs:
        c = packed_fp32_forms(o, arch)
        if c is None:
            if os.path.exists(os.path.splitext(o)[0] + '.hip'):
                raise CheckError(f'{o}: built from a .hip source but holds no device code')
            continue
        report[os.path.basename(o)] = dict((f'{k[0]} {k[1]}', v) for k, v in sorted(c.items()))
        unsafe += sum(v for k, v in c.items() if k[1] == 'op_sel')
    return unsafe, report


if __name__ == '__main__':
    argv = sys.argv[1:]
    arch = 'gfx950'
    if '--arch' in argv:
        i = argv.index('--arch')
        arch = argv[i + 1]
        del argv[i:i + 2]
    try:
        unsafe, report = check(argv or None, arch)
    except CheckError as ex:
        print('check_packed_fp32: the check could NOT be made:', ex)
        sys.exit(2)
    for o, r in report.items():
        print(o, r)
    print(f'unsafe packed fp32 instructions (a high source half in the low lane), {arch}:', unsafe)
    sys.exit(1 if unsafe else 0)
