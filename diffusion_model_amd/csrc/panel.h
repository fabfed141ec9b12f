// Internal interface between dm_gemm (gemm.hip) and the row-panel kernels (panel.hip).
#pragma once
#include "dm_hip.h"

// rows per block of a DM_TILE_P* tile id, 0 for every other tile
int dm_panel_rows(int tile);
// 1 when the problem can run on the panel tile with `nsplit` column slices (grid.y)
int dm_panel_supported(const DmGemmDesc* d, int tile, int nsplit);
int dm_panel_launch(const DmGemmDesc* d, int tile, int nsplit, hipStream_t stream);
