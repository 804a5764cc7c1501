#pragma once
#include <hipcub/hipcub.hpp>
