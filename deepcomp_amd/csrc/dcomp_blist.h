// Base-station counts the step/reset kernels are instantiated for (one object file each): every B the 32-bit
// connection mask can hold.  deepcomp_amd/build.py reads this list; keep the two macros in sync.
#pragma once
#ifndef DCOMP_B_LIST
#define DCOMP_B_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#define DCOMP_B_LIST_STR "1-32"
#endif
