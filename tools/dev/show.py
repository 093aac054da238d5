"""Development aid: headline and per-stage convolution times of a bench log."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d['gpu_time']['stages_ms_per_step']
print(sys.argv[2] if len(sys.argv) > 2 else '', round(d['value'], 1), 'pages/s', round(d['gpu_time']['kernels_ms_per_step'], 2), 'ms',
      {k.replace('conv_wino4', 'w4').replace(',tile16x32', '').replace(',tile8x32', ''): v for k, v in st.items() if 'wino4' in k})
