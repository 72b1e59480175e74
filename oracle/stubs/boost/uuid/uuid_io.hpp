#pragma once
#include <boost/uuid/uuid_generators.hpp>
