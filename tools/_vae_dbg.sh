cd $GRAFT_REPO_ROOT
export TDG_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in a b; do
python3 tests/_dist_worker.py /tmp/t1$mode.npz vae towers
python3 tests/_dist_worker.py /tmp/t2$mode.npz vae towers
done
python3 tests/_dist_worker.py /tmp/o1.npz vae same
python3 tests/_dist_worker.py /tmp/o2.npz vae same
python3 - <<PY
import numpy as np
def cmp(a,b):
    A,B=np.load('/tmp/%s.npz'%a),np.load('/tmp/%s.npz'%b)
    bad=[k for k in A.files if not np.array_equal(A[k],B[k])]
    print(a,b,len(bad),'of',len(A.files), bad[:4], [A[k][0] for k in A.files if k.startswith('loss_decoder')], [B[k][0] for k in B.files if k.startswith('loss_decoder')])
cmp('t1a','t2a'); cmp('t1b','t2b'); cmp('t1a','t1b'); cmp('o1','o2')
PY
