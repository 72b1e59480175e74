// Build-only stand-in for <nlohmann/json.hpp>: config.h only names the type. TEST INFRASTRUCTURE.
#pragma once
namespace nlohmann {
class json {};
}  // namespace nlohmann
