"""Developer tool: static profile of one kernel in the assembly listing (make -C csrc asm): the
basic blocks with the most instructions and what they are made of.
usage: isa_blocks.py <listing.s> <substring of the mangled kernel name> [min block size]"""
import re, sys
src = open(sys.argv[1]).read()
key = sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 25
m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):', src, re.M)
start = m.end()
end = src.index('.Lfunc_end', start)
blocks, cur = [], ['entry', []]
for ln in src[start:end].split('\n'):
    t = ln.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur)
        cur = [t.split(':')[0], []]
        continue
    if not t or t[0] in ';.':
        continue
    cur[1].append(t)
blocks.append(cur)
count = lambda ins, pat: sum(1 for i in ins if re.match(pat, i))
print(len(blocks), 'blocks,', sum(len(b[1]) for b in blocks), 'instructions,', sum(count(b[1], r'v_') for b in blocks), 'VALU')
for name, ins in blocks:
    if len(ins) >= least:
        print(f"{name:12s} n={len(ins):4d} valu={count(ins, 'v_'):4d} f64={count(ins, r'v_(mul|add|fma)_f64'):3d} "
              f"div={count(ins, 'v_div_scale') // 2} sqrt={count(ins, 'v_rsq_f64')} cmp={count(ins, 'v_cmp')} cnd={count(ins, 'v_cndmask')} "
              f"mov={count(ins, 'v_mov|v_accvgpr')} readlane={count(ins, 'v_readlane')} writelane={count(ins, 'v_writelane')} "
              f"sload={count(ins, 's_load')} ds={count(ins, 'ds_')} scratch={count(ins, 'scratch_')} salu={count(ins, 's_') - count(ins, 's_load')}")
