"""Static resources and main-loop instruction mix of the matrix-core kernels, from the compiler's own assembly (no GPU needed).

    python scripts/isa_report.py [gemm_nt_f16 gemm_nt_h3 gemm_tn ...] > profiles/rNN_isa_resources.txt

For every translation unit named (default: the three that carry the update - gemm_nt_f16, gemm_nt_h3, gemm_tn) the script asks
hipcc for the gfx950 device assembly (`--cuda-device-only -S`, the flags of ase_amd/csrc/Makefile), reads each kernel's
`.amdhsa` / metadata block (VGPRs, AGPRs, SGPRs, LDS, scratch, spills -> waves per SIMD) and, for kernels with a matrix loop,
the instruction mix of the basic block that holds the most MFMAs (the steady-state K-loop body): MFMAs, LDS reads, LDS-DMA
(global_load_lds) / global loads, barriers, waits, VALU, SALU.  What the rocprofv3 counters under profiles/ measure at run
time (MFMA busy, LDS conflicts) reads against these counts."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ase_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '--cuda-device-only', '-S']


def demangle(names):
    """Kernel symbol -> name<template arguments> (binutils' c++filt does not know the _Float16 / __bf16 manglings DF16_ / DF16b)."""
    out = []
    for sym in names:
        m = re.match(r'^_ZN12_GLOBAL__N_1(\d+)', sym)
        if not m:
            out.append(sym)
            continue
        n = int(m.group(1))
        name, rest = sym[m.end():m.end() + n], sym[m.end() + n:]
        args = []
        if rest.startswith('I'):
            rest = rest[1:]
            while rest and not rest.startswith('E'):
                for pat, fn in ((r'^DF16_', lambda g: 'f16'), (r'^DF16b', lambda g: 'bf16'), (r'^Li(\d+)E', lambda g: g.group(1)),
                                (r'^Lb([01])E', lambda g: 'true' if g.group(1) == '1' else 'false'), (r'^f', lambda g: 'float'),
                                (r'^(\d+)', None)):
                    g = re.match(pat, rest)
                    if not g:
                        continue
                    if fn is None:                       # <length><identifier>
                        k = int(g.group(1))
                        args.append(rest[g.end():g.end() + k])
                        rest = rest[g.end() + k:]
                    else:
                        args.append(fn(g))
                        rest = rest[g.end():]
                    break
                else:
                    args.append('?')
                    break
        out.append(name + ('<' + ', '.join(args) + '>' if args else ''))
    return out


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'lds_read'
    if op.startswith('ds_'):
        return 'lds_other'
    if op.startswith('global_load_lds') or (op.startswith('buffer_load') and 'lds' in op):
        return 'lds_dma'
    if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'):
        return 'global_load'
    if op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('global_atomic'):
        return 'global_store'
    if op == 's_barrier':
        return 'barrier'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_setprio') or op.startswith('s_nop') or op.startswith('s_sleep'):
        return 'sched'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def parse(asm):
    """-> {kernel symbol: {'meta': {...}, 'blocks': [[ops...], ...]}}"""
    kernels = collections.OrderedDict()
    cur, block = None, None
    for line in asm.split('\n'):
        s = line.strip()
        m = re.match(r'^([A-Za-z_][\w.$]*):', s)
        if m and not s.startswith('.L'):
            name = m.group(1)
            cur = kernels.setdefault(name, {'meta': {}, 'blocks': [[]]})
            block = cur['blocks'][-1]
            continue
        if cur is None:
            continue
        if s.startswith('.LBB') and s.endswith(':') or re.match(r'^\.LBB[\w]*:', s):
            cur['blocks'].append([])
            block = cur['blocks'][-1]
            continue
        if s.startswith('.end_amdhsa_kernel') or s.startswith('.section') or s.startswith('.text'):
            continue
        if not s or s.startswith(';') or s.startswith('.') or s.startswith('//'):
            continue
        op = s.split()[0]
        if re.match(r'^[a-z][a-z0-9_]+$', op):
            block.append(op)
    # metadata (YAML at the end of the file): one entry per kernel
    for ent in re.split(r'\n\s*- \.agpr_count:', asm)[1:]:
        ent = '.agpr_count:' + ent
        nm = re.search(r'\.symbol:\s+(\S+)\.kd', ent)
        if not nm or nm.group(1) not in kernels:
            continue
        meta = kernels[nm.group(1)]['meta']
        for key in ('agpr_count', 'vgpr_count', 'sgpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size',
                    'vgpr_spill_count', 'sgpr_spill_count', 'max_flat_workgroup_size'):
            v = re.search(r'\.' + key + r':\s+(\d+)', ent)
            if v:
                meta[key] = int(v.group(1))
    return kernels


def waves_per_simd(meta):
    regs = meta.get('vgpr_count', 0) + meta.get('agpr_count', 0)       # gfx950: unified 512-entry file per SIMD lane
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def report(tu, out):
    src = os.path.join(CSRC, tu + '.hip')
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, tu + '.s')
        subprocess.run(['hipcc'] + FLAGS + [src, '-o', dst], check=True, cwd=CSRC)
        asm = open(dst).read()
    kernels = {k: v for k, v in parse(asm).items() if v['meta']}
    names = list(kernels)
    pretty = dict(zip(names, demangle(names)))
    out.write(f'== {tu}.hip  ({len(kernels)} kernels)\n')
    out.write(f'{"VGPR":>5} {"AGPR":>5} {"SGPR":>5} {"scratch B":>9} {"spills":>6} {"waves/SIMD":>10}  {"MFMA (total)":>12}  kernel\n')
    rows = []
    for k, v in kernels.items():
        m = v['meta']
        total = collections.Counter(classify(op) for b in v['blocks'] for op in b)
        rows.append((total['mfma'], k, m, total))
    for n_mfma, k, m, total in sorted(rows, key=lambda r: -r[0]):
        nm = re.sub(r'\(anonymous namespace\)::', '', pretty[k])
        nm = re.sub(r'^void ', '', nm).replace('(ase_nt::NTParams)', '').replace('_Float16', 'f16').replace('__bf16', 'bf16')
        out.write(f'{m.get("vgpr_count", 0):>5} {m.get("agpr_count", 0):>5} {m.get("sgpr_count", 0):>5} '
                  f'{m.get("private_segment_fixed_size", 0):>9} {m.get("vgpr_spill_count", 0) + m.get("sgpr_spill_count", 0):>6} '
                  f'{waves_per_simd(m):>10}  {n_mfma:>12}  {nm[:150]}\n')
    out.write('\n   steady-state loop body = the basic block with the most MFMAs, per kernel with >= 8 of them in one block:\n')
    keys = ['mfma', 'lds_read', 'lds_dma', 'global_load', 'lds_other', 'barrier', 'waitcnt', 'sched', 'valu', 'salu', 'branch']
    out.write('   ' + ' '.join(f'{k:>11}' for k in keys) + '   kernel\n')
    for n_mfma, k, m, total in sorted(rows, key=lambda r: -r[0]):
        best = max(kernels[k]['blocks'], key=lambda b: sum(1 for op in b if classify(op) == 'mfma'))
        c = collections.Counter(classify(op) for op in best)
        if c['mfma'] < 8:
            continue
        nm = re.sub(r'\(anonymous namespace\)::', '', pretty[k])
        nm = re.sub(r'^void ', '', nm).replace('(ase_nt::NTParams)', '').replace('_Float16', 'f16').replace('__bf16', 'bf16')
        out.write('   ' + ' '.join(f'{c[x]:>11}' for x in keys) + f'   {nm[:120]}\n')
    out.write('\n')


def main():
    tus = sys.argv[1:] or ['gemm_nt_f16', 'gemm_nt_h3', 'gemm_tn']
    sys.stdout.write('Static kernel resources, gfx950, hipcc ' + ' '.join(FLAGS[:-2]) + '\n'
                     '(VGPR + AGPR share a 512-entry file per SIMD lane: waves/SIMD = 512 // registers, capped at 8; a 512-thread workgroup\n'
                     ' needs 2 per SIMD; "scratch" = private segment bytes per lane - local arrays of the epilogues, spills are their own column)\n\n')
    for tu in tus:
        report(tu, sys.stdout)


if __name__ == '__main__':
    main()
