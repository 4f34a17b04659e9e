// oracle/ref_clustering_shim.cc -- TEST INFRASTRUCTURE ONLY.
// extern "C" door onto the REFERENCE's own L3D::performClustering (clustering.cc:6-47,
// universe.h:59-115), compiled from /root/reference in place by oracle/Makefile into
// oracle/_ref/libclustering_ref.so.  Used to pin oracle/l3d_oracle.c:l3do_clustering and
// the product's host clustering.  This file contains no reference text.
#include <list>
#include "clustering.h"

extern "C" int l3dref_clustering(const int* ei, const int* ej, const float* ew, int E,
                                 int numNodes, float c, int* labels)
{
    std::list<L3D::CLEdge> edges;
    for (int k = 0; k < E; ++k) {
        L3D::CLEdge e;
        e.i_ = ei[k]; e.j_ = ej[k]; e.w_ = ew[k];
        edges.push_back(e);
    }
    L3D::CLUniverse* u = L3D::performClustering(edges, numNodes, c);
    if (u == NULL)
        return 1;
    for (int k = 0; k < numNodes; ++k)
        labels[k] = u->find(k);
    delete u;
    return 0;
}

// The orders SparseMatrix gives its entries (sparsematrix.cc:78-84: `entries.sort(L3D::sortCLEdgesByRow / ByCol)` on a std::list, the
// comparators of clustering.h:98-116): pins the oracle's stable_sort_edges.  In place.
extern "C" void l3dref_sort_cledges(int* ei, int* ej, float* ew, int E, int by_row)
{
    std::list<L3D::CLEdge> entries;
    for (int k = 0; k < E; ++k) {
        L3D::CLEdge e;
        e.i_ = ei[k]; e.j_ = ej[k]; e.w_ = ew[k];
        entries.push_back(e);
    }
    if (by_row) entries.sort(L3D::sortCLEdgesByRow);
    else entries.sort(L3D::sortCLEdgesByCol);
    int k = 0;
    for (std::list<L3D::CLEdge>::const_iterator it = entries.begin(); it != entries.end(); ++it, ++k) { ei[k] = it->i_; ej[k] = it->j_; ew[k] = it->w_; }
}
