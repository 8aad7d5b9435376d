#!/usr/bin/env python3
"""Where does a streamed-weight kernel touch vector memory OUTSIDE its LDS-DMA weight stream?  (CPU; reads the shipped object.)

The stream (ra_stream.hpp Pipe) keeps 12 LDS-DMA pieces in flight behind COUNTED `s_waitcnt vmcnt(N)`.  Every other VMEM operation of the
wave shares that counter: a compiler-visible load comes with the compiler's own `vmcnt(k)`, computed without knowledge of the asm DMA, i.e.
it drains the stream down to k operations; a scratch reload (spill) comes with `vmcnt(0)`.  This tool disassembles one object file and
lists, per kernel: registers / spills / LDS, and every scratch access, plain global load / store and `vmcnt(0)` with its position in the
tile (number of MFMAs before it), plus the histogram of the counted waits.

    python3 tools/isa_vmem.py relightableavatar_amd/csrc/ra_k4_bwd_f16.o [kernel-name substring]"""
import bisect, os, re, subprocess, sys, tempfile
from collections import Counter

LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    obj = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'dev.co')
        subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', obj], check=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--output={co}'], check=True)
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
        asm = subprocess.run([f'{LLVM}/llvm-objdump', '-d', co], capture_output=True, text=True).stdout.split('\n')
    meta, cur = {}, {}
    for l in notes.split('\n'):
        m = re.search(r'\.(name|vgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)', l)
        if m:
            if m.group(1) == 'agpr_count' and cur.get('name'):
                meta[cur['name']] = cur
                cur = {}
            cur[m.group(1)] = m.group(2)
    if cur.get('name'):
        meta[cur['name']] = cur
    starts = [(i, l) for i, l in enumerate(asm) if re.match(r'^[0-9a-f]+ <.*>:', l)]
    for k, (i, l) in enumerate(starts):
        name = re.search(r'<(.*)>', l).group(1)
        if pat not in name:
            continue
        body = asm[i:(starts[k + 1][0] if k + 1 < len(starts) else len(asm))]
        mf = [j for j, x in enumerate(body) if 'v_mfma' in x]
        if not mf:
            continue
        md = meta.get(name, {})
        print(f'== {name[:110]}')
        print(f'   {len(mf)} MFMAs per tile, vgpr {md.get("vgpr_count")} agpr {md.get("agpr_count")} spilled {md.get("vgpr_spill_count")} '
              f'scratch {md.get("private_segment_fixed_size")} B, LDS {md.get("group_segment_fixed_size")} B')
        at = lambda j: bisect.bisect_left(mf, j)
        inloop = lambda j: 0 < at(j) < len(mf)
        ev = {'scratch_load': [], 'scratch_store': [], 'global_load': [], 'global_store': [], 'vmcnt(0)': []}
        for j, x in enumerate(body):
            if not inloop(j):
                continue
            if 'scratch_load' in x: ev['scratch_load'].append(at(j))
            elif 'scratch_store' in x: ev['scratch_store'].append(at(j))
            elif 'global_load' in x and 'lds' not in x: ev['global_load'].append(at(j))
            elif 'global_store' in x: ev['global_store'].append(at(j))
            elif 's_waitcnt' in x and 'vmcnt(0)' in x: ev['vmcnt(0)'].append(at(j))
        for key, v in ev.items():
            pos = ' '.join(map(str, v)) if len(v) <= 40 else ' '.join(map(str, v[:40])) + ' ...'
            print(f'   {key:14s} inside the tile: {len(v):4d}   (MFMAs before each: {pos})')
        c = Counter(int(re.search(r'vmcnt\((\d+)\)', x).group(1)) for j, x in enumerate(body) if 's_waitcnt' in x and 'vmcnt' in x and inloop(j))
        print('   counted waits inside the tile (vmcnt value: instances): ' + ', '.join(f'{k}: {v}' for k, v in sorted(c.items())))
        print(f'   LDS-DMA pieces (global_load_lds): {sum("global_load_lds" in x for x in body)}')


if __name__ == '__main__':
    main()
