// rpn_knobs.h -- environment knobs of librpn_hip.so: the product set and the laboratory set.
#pragma once
#include <cstdlib>

namespace rpn {
inline int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
}  // namespace rpn

// Environment knobs, read once per process (callers keep the value in a function-local static).
// RPN_KNOB: PRODUCT knobs -- every setting computes the same arithmetic contract (bit-exact integer outputs, floats within
//   the documented bound); they are listed in include/rpn_hip.h and tests/test_host.py checks that the shipped library
//   contains no other RPN_* name.
// RPN_LAB_KNOB: LABORATORY knobs (kernel / tile selection for A/B timing, and timing experiments whose results may be
//   wrong).  They exist only in a -DRPN_LAB build (`make lab` -> librpn_hip_lab.so, selected with RPN_HIP_LIB by the
//   scripts/*_ab.sh tooling); in the product library the macro is its default and the name is not even in the binary.
#define RPN_KNOB(name, dflt) rpn::env_int(name, dflt)
#ifdef RPN_LAB
#define RPN_LAB_KNOB(name, dflt) rpn::env_int(name, dflt)
#else
#define RPN_LAB_KNOB(name, dflt) (dflt)
#endif
