/*
 * compat/dubins.h -- the part of the dubins_curves C API that the afb2001/mpc host code touches
 * (test/test_controller.cpp:458-463, src/mpc_node.cpp:237-244): the DubinsPath record and the word enum.
 * dubins_curves is a catkin workspace dependency of the reference (CMakeLists.txt:19) and is not in this
 * repository; in a real workspace its own header is found first and this file is not used.
 */
#ifndef MPC_AMD_COMPAT_DUBINS_H
#define MPC_AMD_COMPAT_DUBINS_H

typedef enum { LSL = 0, LSR = 1, RSL = 2, RSR = 3, RLR = 4, LRL = 5 } DubinsPathType;

typedef struct {
    double qi[3];        /* start pose: x, y, yaw (mathematical: 0 = +x, counter-clockwise) */
    double param[3];     /* lengths of the three segments in units of rho */
    double rho;          /* turning radius */
    DubinsPathType type;
} DubinsPath;

#endif
