// Build-only stand-in for boost uuid (used by utils.cpp:25-30 generateRandomHash, not on the hot path).
#pragma once
#include <string>
namespace boost { namespace uuids {
struct uuid {};
struct random_generator { uuid operator()() { return uuid(); } };
inline std::string to_string(const uuid&) { return "00000000-0000-0000-0000-000000000000"; }
}}  // namespace boost::uuids
