//! The star network on one multi-GPU node: `zk_net` behind the reference's `MpcNet` surface
//! (`mpc-net/src/lib.rs:43-53, 60-176`; `ser_net.rs:16-120`).  One process per GPU; a rank drives `k = n / world`
//! parties, rank 0 does the king's work; gather / scatter run over RCCL (xGMI) on device buffers.
use core::ffi::c_void;
use core::ptr;

use mpc_net::{MpcNetError, MultiplexedStreamID};
use zksaas_hip_sys as sys;

use crate::{check, Context};

/// What the primitives need from a net: the context and the `zk_net` of this rank.  `HipNet` implements it; a host
/// that already has an `MpcNet` implementation wraps one beside it.
pub trait HipBacked {
    fn ctx(&self) -> &Context;
    fn raw_net(&self) -> *mut sys::ZkNet;
    /// parties driven by this rank (1 in the reference's one-process-per-party deployment)
    fn parties_per_rank(&self) -> usize;
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Transport {
    Local,
    Rccl,
    Shm,
}

pub struct HipNet {
    ctx: Context,
    raw: *mut sys::ZkNet,
    pub rank: usize,
    pub world: usize,
    pub first_party: usize,
    pub k: usize,
}
unsafe impl Send for HipNet {}
unsafe impl Sync for HipNet {}

impl HipNet {
    /// The 512-byte id rank 0 makes and the launcher hands to every rank (`zk_net_unique_id`).
    pub fn unique_id() -> Result<Vec<u8>, MpcNetError> {
        let mut id = vec![0u8; sys::ZK_NET_ID_BYTES];
        let rc = unsafe { sys::zk_net_unique_id(id.as_mut_ptr() as *mut c_void) };
        if rc != sys::ZK_OK {
            return Err(MpcNetError::Generic(format!("zk_net_unique_id failed ({rc})")));
        }
        Ok(id)
    }
    pub fn new(ctx: &Context, transport: Transport, rank: usize, world: usize, id: &[u8],
               party_to_rank: Option<&[i32]>) -> Result<Self, MpcNetError> {
        let t = match transport {
            Transport::Local => sys::ZK_NET_LOCAL,
            Transport::Rccl => sys::ZK_NET_RCCL,
            Transport::Shm => sys::ZK_NET_SHM,
        };
        let mut raw: *mut sys::ZkNet = ptr::null_mut();
        let map = party_to_rank.map(|m| m.as_ptr()).unwrap_or(ptr::null());
        check(ctx, unsafe {
            sys::zk_net_create(ctx.raw(), t, rank as i32, world as i32, ctx.n as i32, map, id.as_ptr() as *const c_void, 0,
                               &mut raw)
        })?;
        let mut info = [0i32; 4];
        check(ctx, unsafe { sys::zk_net_info(raw, info.as_mut_ptr()) })?;
        Ok(HipNet { ctx: ctx.clone(), raw, rank, world, first_party: info[2] as usize, k: info[3] as usize })
    }
    /// `MpcNet::is_king` (`lib.rs:65-67`): party 0 lives on rank 0.
    pub fn is_king(&self) -> bool {
        self.rank == 0
    }
    /// `MpcNet::n_parties` (`lib.rs:69`).
    pub fn n_parties(&self) -> usize {
        self.ctx.n
    }
    /// `MpcNet::party_id` (`lib.rs:71`): the first party this rank drives.
    pub fn party_id(&self) -> u32 {
        self.first_party as u32
    }
    /// The round timeout (`lib.rs:98-135`, 30 s by default there and here).
    pub fn set_timeout_ms(&self, ms: u64) -> Result<(), MpcNetError> {
        check(&self.ctx, unsafe { sys::zk_net_set_timeout_ms(self.raw, ms) })
    }
    /// Waits for a channel's transfers with the timeout as a deadline (a hung collective becomes `Protocol`).
    pub fn sync(&self, sid: MultiplexedStreamID) -> Result<(), MpcNetError> {
        check(&self.ctx, unsafe { sys::zk_net_sync(self.raw, sid as i32) })
    }
}
impl Drop for HipNet {
    fn drop(&mut self) {
        unsafe { sys::zk_net_destroy(self.raw) }
    }
}
impl HipBacked for HipNet {
    fn ctx(&self) -> &Context {
        &self.ctx
    }
    fn raw_net(&self) -> *mut sys::ZkNet {
        self.raw
    }
    fn parties_per_rank(&self) -> usize {
        self.k
    }
}
