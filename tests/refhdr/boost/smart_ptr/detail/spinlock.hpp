// stand-in (declarations only): what core/common/Lock.h touches
#pragma once
namespace boost {
namespace detail {
void yield(unsigned k);
}
}  // namespace boost
