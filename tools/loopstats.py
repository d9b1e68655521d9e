import re,collections,sys
src=sys.argv[1] if len(sys.argv)>1 else '/tmp/fa_build/k128.s'
raw=[l.rstrip('\n') for l in open(src)]
# loop header label
start=None
for i,l in enumerate(raw):
    if re.match(r'^\.LBB\d+_\d+:.*Loop Header',l) and start is None:
        # pick the loop that contains >=100 mfma
        lab=l.split(':')[0]
        for j in range(i+1,len(raw)):
            if re.search(r's_cbranch_\w+\s+'+re.escape(lab)+r'\b',raw[j]):
                n=sum('v_mfma' in x for x in raw[i:j])
                if n>=100: start,end=i,j
                break
lines=[l.strip() for l in raw[start:end+1] if l.strip() and not l.strip().startswith(';')]
ops=collections.Counter(l.split()[0] for l in lines if not l.startswith('.L'))
print('loop lines',len(lines),'mfma',ops['v_mfma_f32_16x16x32_f16'])
print(ops.most_common(50))
gaps=[];g=[]
for l in lines:
    if l.startswith('v_mfma'):
        gaps.append(g); g=[]
    elif not l.startswith('.L'):
        g.append(l.split()[0])
print([len(x) for x in gaps])
if len(sys.argv)>2:
    for k,x in enumerate(gaps[:70]): print(k,' '.join(x))

COST={'v_exp_f32_e32':9,'v_cvt_pk_f16_f32':7.8,'v_pk_add_f16':5.4,'v_dot2c_f32_f16_e32':8,'v_max3_f32':5,'v_max_f32_e32':5,
'ds_read_b128':7.5,'ds_read_b64_tr_b16':7.5,'s_waitcnt':4,'v_add_u32_e32':5,'s_mov_b32':4,'s_nop':4,'buffer_load_dwordx4':8,
'v_mov_b32_e32':5,'v_cmp_lt_f32_e32':8,'v_add_f32_e32':5,'v_fma_f32':5}
tot=0; ideal=0
costs=[]
for x in gaps[1:]+[g]:
    F=sum(COST.get(i,4) for i in x)
    costs.append(round(F))
    tot+=max(32,8+F); ideal+=8+F
n=len(gaps)
print('per-gap filler cost',costs)
print('sum max(32,8+F) per step (6 steps):',tot/6,' pure issue per step:',ideal/6)
