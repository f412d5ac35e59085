"""One-off check of PixelTable.from_cool and open_matrix's h5py route (h5py is only in the conda interpreter of the
build image): writes a synthetic table in cooler's HDF5 layout and reads it back.
    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tools/check_cool_reader.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, h5py
from stripenn_amd import pixels, synth
names=['chrA','chrB']; chroms={'chrA':synth.SynthChrom(900,41),'chrB':synth.SynthChrom(700,42)}
t=pixels.PixelTable.from_synth(names,chroms,5000)
path='/tmp/t.mcool'
with h5py.File(path,'w') as f:
    g=f.create_group('resolutions/5000')
    g.attrs['bin-size']=5000
    g.create_dataset('chroms/name',data=np.array(names,dtype='S'))
    g.create_dataset('chroms/length',data=t.chromsizes)
    nb=int(t.chrom_offset[-1])
    chrom_id=np.repeat(np.arange(2),np.diff(t.chrom_offset))
    start=np.concatenate([np.arange(n)*5000 for n in np.diff(t.chrom_offset)])
    g.create_dataset('bins/chrom',data=chrom_id); g.create_dataset('bins/start',data=start); g.create_dataset('bins/end',data=start+5000)
    g.create_dataset('bins/weight',data=t.weights['weight'])
    kr=1.0/np.where(np.isnan(t.weights['weight']),np.nan,t.weights['weight']*1.37)      # a divisive column as hic2cool writes it
    g.create_dataset('bins/KR',data=kr)
    g.create_dataset('pixels/bin1_id',data=t.bin1_id); g.create_dataset('pixels/bin2_id',data=t.bin2_id); g.create_dataset('pixels/count',data=t.count)
    g.create_dataset('indexes/chrom_offset',data=t.chrom_offset)
u=pixels.PixelTable.from_cool(path,'resolutions/5000')
assert u.chromnames==names and u.binsize==5000
for a in ('chromsizes','chrom_offset','bin1_id','bin2_id','count'): assert np.array_equal(getattr(u,a),getattr(t,a)),a
assert np.array_equal(u.weights['weight'],t.weights['weight'],equal_nan=True)
from stripenn_amd import io
sys.modules['cooler']=None          # force the h5py route of open_matrix
info=io.open_matrix(path+'::resolutions/5000')
print(info.chromnames, info.binsize, list(info.bins().columns), info.matrix(balance='weight').fetch('chrB').shape)
# the divisive KR column (cooler: bias = 1 / KR) through the h5py route: hand-computed on a window
selkr=info.matrix(balance='KR')
blk=selkr.fetch('chrA:500001-1000000')
cnt=chroms['chrA'].counts(100,200,100,200)
b=1.0/kr[100:200]
with np.errstate(invalid='ignore'):
    exp=np.where(cnt>0, cnt*np.outer(b,b), 0.0)
assert np.array_equal(blk,exp,equal_nan=True)
# float counts survive the file
with h5py.File(path,'a') as f:
    del f['resolutions/5000/pixels/count']
    f['resolutions/5000'].create_dataset('pixels/count',data=t.count*0.25)
v=pixels.PixelTable.from_cool(path,'resolutions/5000')
assert v.count.dtype==np.float64 and np.array_equal(v.count,t.count*0.25)
print('from_cool round trip ok (weight, divisive KR column, float counts)')
