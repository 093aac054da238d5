"""Developer probe: does the process exit cleanly under rocprofv3 with / without the cooperative LSTM launch?"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from ocr_amd.engine import HipContext
from ocr_amd.weights import random_crnn_params
ctx = HipContext(0)
ctx.crnn_load_weights(random_crnn_params(97, seed=2))
ctx.set_option('lstm_persistent', int(sys.argv[1]))
x = torch.rand((4, 64, 128), device='cuda')
ctx.crnn_forward(x, want_logits=True)
torch.cuda.synchronize()
ctx.close()
print('done', sys.argv[1])
