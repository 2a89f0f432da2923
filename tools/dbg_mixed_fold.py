import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from golden_util import load_deck, rel_err
from oracle_lib import Oracle
from unconfined_amd.abi import params_from_deck
from unconfined_amd import engine
o=Oracle()
for name in ["hantush_lay2", "neuman74_partpen", "c3_moench", "hstorage_partpen_lay2", "mishra_fd30"]:
    for which in ("top","bottom"):
        dk,ts,P0=load_deck(name)
        dk = dk.replace(d=0.0) if which=="top" else dk.replace(l=dk.b)
        P=params_from_deck(dk); D=o.nondim(P)
        zmid=0.5*(dk.d+dk.l); zD=np.array([zmid, 0.03*dk.b if which=="bottom" else 0.97*dk.b])/D.Lc
        zl=np.asarray(o.zlay(D,zD)); keep=zl!=3; zD,zl=zD[keep],zl[keep]
        rng=np.random.default_rng(17); n=320
        tD=10.0**rng.uniform(-1,4,n); rD=10.0**rng.uniform(-0.5,0.7,n); sv=o.split_vector(list(dk.j0s),tD)
        ho,dho=o.batch(P,tD,rD,sv,zD,zl); floor=1e-3*np.nanmax(np.abs(ho))
        for mode in ("faithful","fast"):
            pl=engine.Plan(P,mode=mode); h,dh=pl.drawdown(tD,rD,sv,zD,zl); pl.close()
            e=rel_err(h,ho,floor).ravel(); ed=rel_err(dh,dho,floor).ravel()
            print(name,which,mode,'h p50 %.1e p99 %.1e max %.1e | dh p50 %.1e p99 %.1e max %.1e'%(np.median(e),np.quantile(e,0.99),e.max(),np.median(ed),np.quantile(ed,0.99),ed.max()), 'nan eq', np.array_equal(np.isnan(h),np.isnan(ho)))
