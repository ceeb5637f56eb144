// Base-station counts the step/reset kernels are instantiated for (one object file each).
// deepcomp_amd/build.py reads this list; keep the two macros in sync.
#pragma once
#ifndef DCOMP_B_LIST
#define DCOMP_B_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(16) X(20) X(24) X(32)
#define DCOMP_B_LIST_STR "1-10,12,16,20,24,32"
#endif
