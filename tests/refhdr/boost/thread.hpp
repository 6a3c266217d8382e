// stand-in (declarations only): what core/common/Lock.h touches
#pragma once
namespace boost {
class shared_mutex {
public:
    void lock();
    void unlock();
    void lock_shared();
    void unlock_shared();
};
template <class M>
class shared_lock {
public:
    explicit shared_lock(M&);
    ~shared_lock();
};
template <class M>
class unique_lock {
public:
    explicit unique_lock(M&);
    ~unique_lock();
};
}  // namespace boost
