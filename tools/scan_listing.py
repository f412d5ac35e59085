"""Scan the device listing of the library for loads that are waited for one at a time.

    python tools/scan_listing.py [kernel-name-substring ...]

Compiles stripenn_amd/csrc/stripenn_hip.hip to gfx950 assembly with the product flags (hipcc -S --cuda-device-only, ~1 min) and
reports, per kernel: registers / scratch / occupancy, and the places where ONE global (or LDS) load is followed within a few
instructions by `s_waitcnt vmcnt(0)` (`lgkmcnt(0)`) -- the signature of a load whose first use sits next to it inside a branch, so
that the compiler could not hoist it and every element pays its own round trip.  Round 5 found the resolver's tap loads
(k_canny_f32), the short rows and tails of k_score_wave and the totals of k_lines this way; a site is a hint, not a verdict
(the last load of a batch also matches): look at the listing around it (--show N prints N lines around every site).
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize',
         '-fvisibility=hidden', '-DSTP_BUILD', '--cuda-device-only', '-S']


def listing():
    out = os.path.join(tempfile.gettempdir(), 'stripenn_hip_gfx950.s')
    src = os.path.join(ROOT, 'stripenn_amd', 'csrc', 'stripenn_hip.hip')
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    show = 0
    if '--show' in sys.argv:
        show = int(sys.argv[sys.argv.index('--show') + 1]); args = [a for a in args if a != str(show)]
    s = listing()
    for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M):
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        short = name.split('(')[0].replace('void ', '')
        if args and not any(a in short for a in args):
            continue
        rest = s[m.end():]
        if '.end_amdhsa_kernel' not in rest:
            continue
        body = rest[:rest.index('.end_amdhsa_kernel')].splitlines()
        g = lambda n: (re.search(r'; %s: (\d+)' % n, rest) or [None, '?'])[1]
        ins = [(i, l.strip()) for i, l in enumerate(body)
               if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.') and not l.strip().endswith(':')]
        sites = {'global': [], 'lds': []}
        for k, (i, l) in enumerate(ins):
            for kind, op, cnt in (('global', 'global_load', 'vmcnt(0)'), ('lds', 'ds_read', 'lgkmcnt(0)')):
                if not l.startswith(op) or (k and ins[k - 1][1].startswith(op)) or (k + 1 < len(ins) and ins[k + 1][1].startswith(op)):
                    continue
                for j in range(k + 1, min(k + 5, len(ins))):
                    if cnt in ins[j][1]:
                        sites[kind].append(i); break
                    if ins[j][1].startswith(op):
                        break
        print('%-40s %5d instr, %3s VGPRs, scratch %s, occupancy %s | single load + wait: %d global, %d LDS'
              % (short[:40], len(ins), g('NumVgprs'), g('ScratchSize'), g('Occupancy'), len(sites['global']), len(sites['lds'])))
        if show:
            for i in sites['global']:
                print('  ---- line %d' % i)
                print('\n'.join('  ' + x[:110] for x in body[max(0, i - show):i + show]))


if __name__ == '__main__':
    main()
