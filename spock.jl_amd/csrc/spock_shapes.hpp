// spock_shapes.hpp -- the compile-time shapes (nx, nu, d, ny) the kernels are instantiated for; every other shape runs the
// run-time-dimension variants.  Experiment builds narrow the list (tools/build_variant.py, -DSPOCK_STATIC_SHAPES=...).
#pragma once
#ifndef SPOCK_STATIC_SHAPES
#define SPOCK_STATIC_SHAPES(X) X(5, 2, 2, 5) X(5, 5, 2, 5) X(2, 1, 2, 5) X(3, 2, 3, 7)
#endif
