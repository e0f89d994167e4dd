#pragma once
// declaration-only stand-in (tests/cpp/thirdparty_stubs/opencv2/core/core.hpp explains)
#include <opencv2/core/core.hpp>
