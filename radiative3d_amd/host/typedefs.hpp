// typedefs.hpp -- scalar / text / exception aliases of the host model-builder.
//
// Source-compatible with the names the reference's user model files rely on
// (reference typedefs.hpp:38-124: Real, Index, Count, Text, TextStream,
// Runtime, Invalid).  fp64 only: the MI355X build fixes Real = double.
#ifndef R3DH_TYPEDEFS_HPP_
#define R3DH_TYPEDEFS_HPP_

#include <sstream>
#include <stdexcept>
#include <string>

using Real = double;

using Index = unsigned int;
using BigIndex = unsigned long;
using SmallIndex = unsigned short;
using RelIndex = int;
using BigRelIndex = long;
using SmallRelIndex = short;
using Count = unsigned int;
using BigCount = unsigned long;
using SmallCount = unsigned short;

using Text = std::string;
using TextStream = std::stringstream;

using Runtime = std::runtime_error;  // the user got something wrong
using Invalid = std::logic_error;    // the programmer got something wrong

#endif
