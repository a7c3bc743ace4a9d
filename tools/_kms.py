import sys,json
d=json.loads(sys.stdin.readline())
k=d['kernels']
print(sys.argv[1] if len(sys.argv)>1 else '', 'value', d['value'], 'stem', k['conv_stem_s2_fused_u8_bf16']['ms'], 'resblock', k['conv_resblock_fused_bf16_64_32_64']['ms'], 'heads', k['conv_head_decode_bf16_128x256']['ms'], 'all', d['roofline']['all_kernels_ms_per_step'])
