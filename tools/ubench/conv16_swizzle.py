import itertools
G1=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; G2=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
def cost(TW, rowb, fn, mtiles=4):
    PW=TW+2; nslot=rowb//16; rows_per_bankrow=256//rowb
    worst=0; total=0; n=0
    for mt in range(mtiles):
      for ky in range(3):
        for kx in range(3):
          for ks in range(nslot//2):
            for half in (0,1):
              for G in (G1,G2):
                pos={}
                for l in G:
                    r=32*mt+l; py,px=r//TW,r%TW
                    prow,pcol=py+ky,px+kx
                    q=prow*PW+pcol
                    logical=2*ks+half
                    phys=logical^fn(prow,pcol)
                    bank=(q%rows_per_bankrow)*nslot+phys     # 16-byte slot within the 256-byte bank row
                    pos[bank]=pos.get(bank,0)+1
                m=max(pos.values()); worst=max(worst,m); total+=m; n+=1
    return worst,total/n
for rowb in (128,64):
    nsl=rowb//16
    for TW in (4,8,16,32):
        best=None
        for a,b,c,d in itertools.product(range(0,4),range(0,8),range(0,4),range(0,8)):
            fn=lambda prow,pcol,a=a,b=b,c=c,d=d:(((pcol>>a)*1+ (prow>>c)*d + b*0)&(nsl-1)) if True else 0
            w,avg=cost(TW,rowb,fn)
            if best is None or (avg,w)<(best[0],best[1]): best=(avg,w,(a,c,d))
        print("rowb",rowb,"TW",TW,"best avg cycles",best)
