"""Build-time guard for the two persistent kernels (deepcv_amd/csrc/resident.hip): parses hipcc's `-Rpass-analysis=kernel-resource-usage` remarks and fails
when `resident_fwd_kernel` / `resident_bwd_kernel` spill VGPRs, use scratch memory, or fall below four waves per SIMD.  Both run one 1024-thread block per CU
(16 waves = four per SIMD, i.e. at most 128 VGPRs); a spilled register is 1 MB of scratch traffic per launch and an exposed L2 round trip at every reload.
usage: check_resources.py <remarks file>   (called by deepcv_amd/csrc/Makefile; prints one summary line per kernel)"""
import re
import sys

KERNELS = ('resident_fwd_kernel', 'resident_bwd_kernel')
# static LDS the kernels may declare: 160 KiB per CU minus the dynamic part of their launch (RS_SMEM = 74240 B, RB_SMEM = 148480 B in resident.hip)
LDS_STATIC_MAX = {'resident_fwd_kernel': 163840 - 74240, 'resident_bwd_kernel': 163840 - 148480}


def parse(text: str):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r'remark:\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def check(text: str):
    """ -> (summary lines, problems) """
    found, lines, problems = parse(text), [], []
    for k in KERNELS:
        hits = [(n, r) for n, r in found.items() if k in n]
        if not hits:
            problems.append(f'{k}: no resource remark found')
            continue
        _, r = hits[0]
        lines.append(f'{k}: VGPRs {r.get("VGPRs")}, VGPR spills {r.get("VGPRs Spill")}, SGPR spills {r.get("SGPRs Spill")}, scratch {r.get("ScratchSize [bytes/lane]")} B/lane, '
                     f'{r.get("Occupancy [waves/SIMD]")} waves/SIMD')
        if r.get('VGPRs Spill', 1) != 0:
            problems.append(f'{k}: {r.get("VGPRs Spill")} VGPR spills')
        if r.get('ScratchSize [bytes/lane]', 1) != 0:
            problems.append(f'{k}: {r.get("ScratchSize [bytes/lane]")} B/lane of scratch memory')
        if r.get('LDS Size [bytes/block]', 1 << 30) > LDS_STATIC_MAX[k]:
            problems.append(f'{k}: {r.get("LDS Size [bytes/block]")} B of static LDS > {LDS_STATIC_MAX[k]} (the launch would fail)')
        if r.get('Occupancy [waves/SIMD]', 0) < 4:
            problems.append(f'{k}: occupancy {r.get("Occupancy [waves/SIMD]")} waves/SIMD < 4 (a 1024-thread block would not fit a CU)')
    return lines, problems


if __name__ == '__main__':
    lines, problems = check(open(sys.argv[1]).read())
    for line in lines:
        print('resources:', line)
    if problems:
        print('resources: FAILED -', '; '.join(problems), file=sys.stderr)
        sys.exit(1)
