import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for sub in ("nmf.jl_amd","oracle","tests"): sys.path.insert(0, os.path.join(R,sub))
import numpy as np, nmfx, nmf_oracle as orc, c_oracle as co
from problems import planted, uniform, rel_trace_err
T=np.float32
os.environ["NMFX_DEV"]="1"
# round 5: the H solve's two routes side by side (strip substitution = the default, product form = NMFX_POTRS=0), three seeds per shape
for shape in [(64,96,5),(300,260,70),(130,515,8),(129,257,100)]:
  for seed_off in (0,1,2):
    p,n,k=shape
    X,W0,H0=planted(p,n,k,T,seed=9+p+100*seed_off,normalize=False,zeroh=True)
    lam=0.05
    o=orc.Opts(maxiter=15,tol=1e-30,lambda_w=lam,lambda_h=lam,track_objective=True)
    r64=orc.solve("projals",np.asfortranarray(X.astype(np.float64)),np.asfortranarray(W0.astype(np.float64)),np.asfortranarray(H0.astype(np.float64)),o)
    kappa=0.0
    W64,H64,X64=np.asfortranarray(W0.astype(np.float64)),np.asfortranarray(H0.astype(np.float64)),np.asfortranarray(X.astype(np.float64))
    for _ in range(15):
        kappa=max(kappa,np.linalg.cond(W64.T@W64+lam*np.eye(k)))
        orc.solve("projals",X64,W64,H64,orc.Opts(maxiter=1,tol=1e-30,lambda_w=lam,lambda_h=lam))
        kappa=max(kappa,np.linalg.cond(H64@H64.T+lam*np.eye(k)))
    res={}
    for route,env in (("strip",None),("product","0")):
        if env is None: os.environ.pop("NMFX_POTRS",None)
        else: os.environ["NMFX_POTRS"]=env
        rr=nmfx.solve(nmfx.ProjectedALS(T,maxiter=15,tol=1e-30,lambda_w=lam,lambda_h=lam),X,W0.copy(order="F"),H0.copy(order="F"),track_objective=True)
        res[route]=rel_trace_err(rr.trace,r64.trace)
    os.environ.pop("NMFX_POTRS",None)
    ro=orc.solve("projals",X,W0.copy(order="F"),H0.copy(order="F"),o)
    rc=co.solve("projals",X,W0.copy(order="F"),H0.copy(order="F"),o)
    print(shape,"seed+%d"%(100*seed_off),"kappa %.2e  kappa*eps %.2e | vs-f64: gpu strip %.2e  gpu product %.2e  numpy32 %.2e  c32 %.2e"%(kappa,kappa*np.finfo(T).eps,res["strip"],res["product"],rel_trace_err(ro.trace,r64.trace),rel_trace_err(rc.trace,r64.trace)),flush=True)
for shape in []:
    p,n,k=shape
    X,W0,H0=planted(p,n,k,T,seed=9+p,normalize=False,zeroh=True)
    lam=0.05
    r=nmfx.solve(nmfx.ProjectedALS(T,maxiter=15,tol=1e-30,lambda_w=lam,lambda_h=lam),X,W0.copy(order="F"),H0.copy(order="F"),track_objective=True)
    o=orc.Opts(maxiter=15,tol=1e-30,lambda_w=lam,lambda_h=lam,track_objective=True)
    ro=orc.solve("projals",X,W0.copy(order="F"),H0.copy(order="F"),o)
    rc=co.solve("projals",X,W0.copy(order="F"),H0.copy(order="F"),o)
    # fp64 ground truth of the same algorithm
    X64,W64,H64=X.astype(np.float64),W0.astype(np.float64),H0.astype(np.float64)
    r64=orc.solve("projals",np.asfortranarray(X64),np.asfortranarray(W64),np.asfortranarray(H64),o)
    print(shape,"gpu-vs-numpy %.2e  c-vs-numpy %.2e  gpu-vs-f64 %.2e  numpy32-vs-f64 %.2e  c32-vs-f64 %.2e"%(rel_trace_err(r.trace,ro.trace),rel_trace_err(rc.trace,ro.trace),rel_trace_err(r.trace,r64.trace),rel_trace_err(ro.trace,r64.trace),rel_trace_err(rc.trace,r64.trace)))
for k in (96,256):
    p,n=640,900
    X,W0,H0=uniform(p,n,k,T,seed=k); lam=0.5
    o=orc.Opts(maxiter=6,tol=1e-30,lambda_w=lam,lambda_h=lam,track_objective=True)
    r=nmfx.solve(nmfx.ProjectedALS(T,maxiter=6,tol=1e-30,lambda_w=lam,lambda_h=lam),X,W0.copy(order="F"),H0.copy(order="F"),track_objective=True)
    ro=orc.solve("projals",X,W0.copy(order="F"),H0.copy(order="F"),o)
    r64=orc.solve("projals",np.asfortranarray(X.astype(np.float64)),np.asfortranarray(W0.astype(np.float64)),np.asfortranarray(H0.astype(np.float64)),o)
    print(k,"gpu-vs-numpy %.2e gpu-vs-f64 %.2e numpy32-vs-f64 %.2e"%(rel_trace_err(r.trace,ro.trace),rel_trace_err(r.trace,r64.trace),rel_trace_err(ro.trace,r64.trace)))
