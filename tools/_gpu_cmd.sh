mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r4_s_pytest.log
tail -6 gpurun_out/r4_s_pytest.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>gpurun_out/r4_s_bench.err | tail -1 > gpurun_out/r4_s_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_s_bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['traffic_source'][:60])
a=j['also']
for k in ('config3_65536x32x10_multi_into_8_fragment_buffers','config3_65536x32x10_multi_resource_fair','config5_share_4096x128x32_multi'):
    print(k, a[k]['kernel_ms'], a[k]['frac_of_hbm_peak'])
print(a['config2_4096x10x5_central_fused_rollout'].get('frac_of_hbm_peak_pmc_traffic'), a['central_65536x10x5']['through_rollout_T50'].get('frac_of_hbm_peak_pmc_traffic'))
"
