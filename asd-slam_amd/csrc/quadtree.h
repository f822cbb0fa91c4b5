// quadtree.h -- host-side DistributeOctTree (reference ORBextractor.cc:587-811).
#pragma once
#include <vector>
// x, y, response: n raw corners of one level in the reference's vToDistributeKeys order
// (coordinates relative to minBorder).  selected receives indices into them, in the reference's
// output order (final node-list order, max response per node, first maximum wins).
void asd_distribute_octtree(const float* x, const float* y, const float* response, int n, int minX, int maxX,
                            int minY, int maxY, int N, std::vector<int>& selected);
