"""K3CC (csrc/ra_k3cc.hpp) addresses its 62 weight-fragment registers a[0:247] BY NAME from inline assembly; the compiler does not know
that a load is pending in them between a `global_load_dwordx4 a[..]` and the counted wait of its k-step.  This check compiles the kernel to
assembly with the product flags and verifies the assumption that makes that safe: outside the inline-asm statements NO instruction writes
an AGPR (the MFMA accumulators are VGPR-form, nothing spills into the AGPR file), and reads of AGPRs happen only after a counted wait.
usage: python tools/check_k3cc_isa.py [path/to/hipcc] | --listing file.s   (exit status 0 = ok; run by csrc/Makefile on the assembly of the
compilation that makes the shipped object — the object is not installed otherwise — and by tests/test_host_logic.py)"""
import os, re, subprocess, sys, tempfile

def compile_listing(hipcc='/opt/rocm/bin/hipcc'):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'relightableavatar_amd', 'csrc', 'ra_k3cc_f16.hip')
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k3cc.s')
        subprocess.run([hipcc, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-value', '-fno-slp-vectorize', '-mllvm',
                        '-amdgpu-mfma-vgpr-form=1', '-S', '--cuda-device-only', src, '-o', out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read().split('\n')


def shipped_listing():
    """the assembly the build kept beside the shipped object (csrc/Makefile: -save-temps of the very compilation that made ra_k3cc_f16.o)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, 'relightableavatar_amd', 'csrc', 'ra_k3cc_f16.s')
    o = os.path.join(root, 'relightableavatar_amd', 'csrc', 'ra_k3cc_f16.o')
    if os.path.exists(p) and os.path.exists(o) and os.path.getmtime(p) <= os.path.getmtime(o) + 60:
        return open(p).read().split('\n')
    return None


def check(hipcc='/opt/rocm/bin/hipcc', lines=None):
    if lines is None:
        lines = shipped_listing() or compile_listing(hipcc)
    in_asm, problems, loads, waits, reads = False, [], 0, 0, 0
    agpr = re.compile(r'\ba(\d+|\[\d+:\d+\])')
    for n, l in enumerate(lines, 1):
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        if in_asm:
            loads += t.startswith('global_load_dwordx4 a[')
            waits += t.startswith('s_waitcnt vmcnt(60)')
            continue
        ops = t.split(None, 1)
        if len(ops) < 2 or not agpr.search(ops[1]):
            continue
        dst = ops[1].split(',')[0].strip()
        if agpr.fullmatch(dst):                                   # an AGPR destination outside inline asm
            problems.append(f'{n}: {t}')
        elif ops[0] == 'v_accvgpr_read_b32' or ops[0].startswith('v_mfma'):
            reads += 1
        else:
            problems.append(f'{n}: {t}')
    if loads != 62 + 992 or waits != 496:
        problems.append(f'expected 1054 fragment loads and 496 counted waits, found {loads} and {waits}')
    return problems, dict(loads=loads, waits=waits, agpr_reads=reads)

if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--listing':        # the build: check the assembly of the compilation that made the object
        problems, stats = check(lines=open(sys.argv[2]).read().split('\n'))
    else:
        problems, stats = check(*sys.argv[1:2])
    print(stats)
    for p in problems[:20]:
        print('PROBLEM', p)
    sys.exit(1 if problems else 0)
