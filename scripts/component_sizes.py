#!/usr/bin/env python3
"""Connected components of the affinity list of a V x 2000 x 12 bench scene (sizes in nodes and edges): the largest one bounds the device
merge loop (k_uf_component walks one component per wave; up to 2048 nodes in LDS).   python scripts/component_sizes.py VIEWS"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
V=int(sys.argv[1])
sc=make_scene(V,2000,12,seed=20260)
l=Line3D("",matchingNeighbors=12); load_scene(l,sc); l.compute3Dmodel(False)
A,n=l.affinity()
nc,lab=connected_components(coo_matrix((np.ones(len(A)),(A["i"],A["j"])),shape=(n,n)).tocsr(),directed=False)
sz=np.bincount(lab); ed=np.bincount(lab[A["i"]],minlength=nc)
o=np.argsort(-ed)[:8]
print("nodes",n,"edges",len(A),"components",nc,"largest by edges:",[(int(sz[k]),int(ed[k])) for k in o], "comps>2048 nodes:", int((sz>2048).sum()))
l.close()
