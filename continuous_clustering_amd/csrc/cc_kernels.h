// cc_kernels.h — the hand-written gfx950 kernels of the continuous-clustering hot path, by stage (all in namespace cck, wave64, compiled with
// -ffp-contract=off: every floating-point expression keeps the reference's operation order, nothing is fused into FMAs the x86 reference does not
// have). One translation unit (cc_engine.hip) includes this file; the order below is the order of definition.
//
//   cc_k_base.h          wave-level helpers: DPP reductions, relaxed LDS accessors, wave-uniform values, per-stream plane pointers
//   cc_k_segcells.h      seg_pre_cells: the per-cell part of the ground segmentation (cc.cpp:294-624), shared by insertion and segmentation
//   cc_k_insert.h        insertFiringIntoRangeImage (cc.cpp:105-292): k_insert_par (block-parallel, fused with the per-cell segmentation and the
//                        inclination-table partials), k_insert_multi (multi-column firings), k_prep + k_insert2 (serial, any input)
//   cc_k_segment.h       k_ego, k_table, k_seg_pre (streams that are not fused), k_seg_scan (table along the columns + row state machine),
//                        k_seg_small (one column by one wavefront, rows as lanes)
//   cc_k_assoc_global.h  scan_point (one point's window scan, cc.cpp:698-771), associate_stream / k_associate (global-memory fallback),
//                        scan_column_epilogue (chains of same-column parents, column summaries, the packed per-column point records)
//   cc_k_scan.h          k_scan / k_scan2 (the window scan of a batch's columns as a pure function of static data), k_small_front
//   cc_k_assoc_lds.h     k_assoc_lds (one-wavefront serial association)
//   cc_assoc_shared.h, cc_assoc3.h   k_assoc3: three / four cooperating wavefronts per stream, the exact serial kernel behind k_assocb
//   cc_assocb.h          k_assocb: batch-parallel association + finished-cluster check (groups of columns, pipelined, points packed into lanes)
//   cc_k_publish.h       k_publish, k_small_tail, frame scatter, cluster gathering, host view
#pragma once
#include <hip/hip_runtime.h>

#include "cc_device.h"
#include "cc_math.h"

#pragma clang fp contract(off)

namespace cck
{
using namespace ccd;

#include "cc_k_base.h"
#include "cc_k_segcells.h"
#include "cc_k_insert.h"
#include "cc_k_segment.h"
#include "cc_k_assoc_global.h"
#include "cc_k_scan.h"
#include "cc_k_assoc_lds.h"
#include "cc_assoc_shared.h"
#include "cc_assoc3.h"
#include "cc_assocb.h"
#include "cc_k_publish.h"

} // namespace cck
