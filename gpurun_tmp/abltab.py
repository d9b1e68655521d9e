import csv,collections,sys
d=sys.argv[1]
print("abl   wave_cyc/step  active  wait_inst  wait_any  mfma_busy  valu_insts/step  active_valu")
for a in sys.argv[2:]:
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f'/root/repo/gpurun_out/{d}/a{a}/p_counter_collection.csv')):
        if 'attn_fwd' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    waves=6144*4; steps=128
    wc=m['SQ_WAVE_CYCLES']*4/waves/steps
    print(f"{a:>4s}  {wc:8.0f}   {m['SQ_ACTIVE_INST_ANY']*4/waves/steps:8.0f} {m['SQ_WAIT_INST_ANY']*4/waves/steps:8.0f} {m['SQ_WAIT_ANY']*4/waves/steps:8.0f} {m['SQ_VALU_MFMA_BUSY_CYCLES']/waves/steps:8.0f} {m['SQ_INSTS_VALU']/waves/steps:8.1f} {m['SQ_ACTIVE_INST_VALU']*4/waves/steps:8.0f}")
