// fhesi_host.h -- C++ mirror of the reference's class surface for the DoubleCRT path, with every body bound to the
// C ABI of include/fhesi_hip.h (the HIP library).  Same class and method names, argument meaning and error behaviour
// (NTL-style Error(msg) -> abort) as the reference so that code written against
//   PAlgebra (PAlgebra.h:53-88), IndexSet (IndexSet.h:26-127), Cmodulus (CModulus.h:42-170), FHEcontext (FHEContext.h:40-205),
//   DoubleCRT (DoubleCRT.h:83-365), CiphertextPart / Ciphertext (Ciphertext.h:10-97), FHESISecKey / FHESIPubKey /
//   KeySwitchSI (FHE-SI.h), Reduce / ReduceCoefficients / DotProduct (Util.h)
// reads the same.  NTL is not available here, so ZZ / ZZX come from zz.h and randomness from a documented SplitMix64
// stream (NTL's SetSeed/RandomBnd and lrand48 streams are not reproducible anyway).
// Rows live in HBM behind fhesi_dcrt handles; getMap()-style access materialises them on demand.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <atomic>
#include <vector>

#include "../../include/fhesi_hip.h"
#include "zz.h"

namespace fhesi {

typedef std::vector<long> vec_long;

inline void ck(int rc) { if (rc) Error(fhesi_last_error()); }

}  // namespace fhesi

// the parts, in dependency order (each mirrors the reference files its header names)
#include "fhesi_numbth.h"       // PRNG, NumbTh.cpp
#include "fhesi_context.h"      // IndexSet, PAlgebra, Cmodulus, FHEcontext
#include "fhesi_doublecrt.h"    // DoubleCRT, SingleCRT
#include "fhesi_util.h"         // samplers, Util.h
#include "fhesi_engine.h"       // device-resident, lazily evaluated ciphertext values (needs FHEcontext above)
#include "fhesi_ciphertext.h"   // Ciphertext, Plaintext
#include "fhesi_keys.h"         // FHESISecKey, FHESIPubKey, KeySwitchSI
