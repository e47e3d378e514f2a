"""Static instruction statistics per kernel of a device ISA listing (scripts/isa.sh <file>): vector-ALU / MFMA / branch / wait counts.
usage: python scripts/isa_stats.py /tmp/isa/<file>.s [name filter]"""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z[A-Za-z0-9_]+:', l)]
dem = subprocess.run(['c++filt'] + [n for _, n in names], capture_output=True, text=True).stdout.split('\n')
for k, (i, n) in enumerate(names):
    j = names[k + 1][0] if k + 1 < len(names) else len(lines)
    body = [x.strip() for x in lines[i:j] if x.startswith('\t') and not x.strip().startswith((';', '.'))]
    if 's_endpgm' in body:
        body = body[:len(body) - body[::-1].index('s_endpgm')]
    name = dem[k].replace('void rpn::', '').split('(')[0]
    if flt not in name:
        continue
    c = lambda p: sum(1 for x in body if x.startswith(p))
    print("%-70s instrs %5d valu %5d mfma %4d salu %5d branches %3d waitcnt %3d barriers %2d lds %4d vmem %3d" % (
        name[:70], len(body), c('v_') - c('v_mfma'), c('v_mfma'), c('s_') - c('s_cbranch') - c('s_waitcnt') - c('s_barrier'), c('s_cbranch'),
        c('s_waitcnt'), c('s_barrier'), c('ds_'), c('buffer_') + c('global_')))
