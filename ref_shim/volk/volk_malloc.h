#pragma once
#include "volk.h"
