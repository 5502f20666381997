import sys, numpy as np
sys.path.insert(0,'/root/repo')
from fullycnnspeechenhancement_amd import FullyCNNTrainer
from oracle import rced_np, train_ref
nw=sys.argv[1] if len(sys.argv)>1 else "FullyCNNV3"
w=rced_np.make_weights(nw,seed=42); x=rced_np.make_input(4,16,seed=1234); y=rced_np.make_input(4,16,seed=1235)
ref=train_ref.TrainRef(nw,w,batch_size=4); lr,gr,_=ref.loss_and_grads(x,y)
tr=FullyCNNTrainer(nw,batch_size=4,lr=1e-3,weights=w); l,_,_=tr.train_step(x,y); g=tr.gradients()
print('loss',l,lr)
for k in gr:
    a=g[k].astype(np.float64); b=gr[k].numpy(); print('%-34s rel %.2e  max|ref| %.3g'%(k, np.abs(a-b).max()/max(np.abs(b).max(),1e-30), np.abs(b).max()))
