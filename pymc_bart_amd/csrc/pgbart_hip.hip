// pgbart_hip.hip -- particle-Gibbs BART for MI355X (gfx950 / CDNA4), C ABI of include/pgbart.h.
//
// Replaces the native sampler the reference imports from the external `bartrs` wheel
// (pymc_bart/pymc_bart.py:2; call site tests/test_bart.py:231-235).  Algorithm: SURVEY.md
// Appendix A (upstream PGBART.astep) under the numeric contract of include/pgbart_spec.h.
//
// Design:
//   * The whole astep is a DEVICE-RESIDENT STATE MACHINE.  The host only enqueues identical
//     "slots" = { k_ctrl ; k_rows [; k_loglik] } on one HIP stream and polls a pinned progress
//     word; it never waits for the device inside a tree update.  k_ctrl (one 256-thread
//     workgroup per particle) finishes the previous SMC round from the integer statistics the
//     row pass produced -- leaf values, particle weights, systematic resampling on one wave, the
//     next growth proposal incl. the exact "k-th row of the leaf" selection -- and writes one
//     job per particle.  k_rows streams the rows on a persistent grid: a work item is a 1024-row
//     chunk times a group of particles with work; it relabels the rows of the leaves being split
//     and reduces the children's sufficient statistics (four values per butterfly, wave_sum4).
//     The slot that starts a tree fuses FINAL(previous tree) + INIT + round 0 into one pass.
//   * HBM layout: X column-major (coalesced column streams), {sum_trees, residual} packed as
//     double2 per row, one BYTE leaf label per row per particle in an 8-generation ring (idle
//     particles are never copied; a pass writes only particles that split).
//   * All row reductions are integer (fixed point) => bit-reproducible, independent of
//     launch geometry and atomics order, and identical to the CPU oracle.
//   * Kernel instances are compiled per data set where the hot loop has no registers or
//     instructions to spare: k_rows<SubsetSplit columns?, Normal family?, linear response?>,
//     k_rows_mk<K>, k_loglik<K, family>, k_ctrl<multi-output?, linear response?, order keys?>.
//   * No MFMA: the path is gather / partition / reduce (HBM / L2 bound).
//
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>  // (device radix sort: the order keys of the design matrix, pgb_set_data)

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <sched.h>
#include <thread>
#include <type_traits>
#include <vector>

#include "pgbart.h"
#include "pgbart_image.h"
#include "pgbart_pack.h"
#include "pgbart_spec.h"

#define CH 1024 /* rows per chunk = rows per k_rows workgroup */
#define BT 256  /* threads per workgroup */
#define RPT (CH / BT)
#define MAXN PGB_MAX_NODES
#define MAXP PGB_MAX_PARTICLES
#define CC_ROUNDS 256
#define NGEN 8 /* generations of particle leaf labels (ring) */


// One translation unit; the parts below are included in this order (each relies on the ones above).
#include "pgb_dev_types.h"
#include "pgb_dev_helpers.h"
#include "pgb_leaf_values.h"
#include "k_ctrl.h"
#include "k_rows.h"
#include "k_rows_mk.h"
#include "k_loglik.h"
#include "k_setup_predict.h"
#include "k_export.h"
#include "pgb_host.h"
#include "pgb_checkpoint.h"
#include "pgb_probe.h"
