"""Per-kernel instruction statistics of a gfx950 assembly listing: MFMA count, scratch (spill) traffic, global loads / stores and
the histogram of s_waitcnt vmcnt(N) values.  A kernel that streams weights with counted waits should show its own N and only a
handful of vmcnt(0); many `vmcnt(0) lgkmcnt(0)` pairs mean flat loads (a pointer that lost its address space).
usage: hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o k.s file.hip; python tools/isa_stats.py k.s [name substring]"""
import re, sys, collections
s=open(sys.argv[1]).read().split('\n')
starts=[(i,l.split(':')[0]) for i,l in enumerate(s) if l.startswith('_ZN12_GLOBAL') and ': ;' in l]
for k,(st,name) in enumerate(starts):
    end=starts[k+1][0] if k+1<len(starts) else len(s)
    if len(sys.argv)>2 and sys.argv[2] not in name: continue
    lines=s[st:end]
    def cnt(p): return sum(1 for l in lines if re.search(p,l))
    vm=[re.search(r'vmcnt\((\d+)\)',l).group(1) for l in lines if re.search(r's_waitcnt.*vmcnt\(',l)]
    print(name[20:75], 'mfma',cnt('v_mfma'),'scr_ld',cnt('scratch_load'),'scr_st',cnt('scratch_store'),'gload',cnt(r'global_load_dword(?!.*lds)'),'gstore',cnt('global_store'), 'accvgpr mov', cnt('v_accvgpr'), 'v_mov', cnt(r'v_mov_b32'), 'vmcnt', sorted(collections.Counter(vm).items(), key=lambda x:int(x[0])))
