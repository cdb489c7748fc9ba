//! The star network on one multi-GPU node as an `mpc_net::MpcNet` (`mpc-net/src/lib.rs:60-176`): one process per GPU,
//! a rank drives `k = n / world` parties (k = 1 with eight GPUs: the reference's one-process-per-party deployment),
//! rank 0 does the king's work; the primitives' gather / scatter run over RCCL (xGMI) on device buffers.
//!
//! `HipNet` implements `MpcNet`, hence `MpcSerNet` (blanket impl, `ser_net.rs:127`), so every function of the
//! reference that is generic over `Net: MpcSerNet` accepts it unchanged.  The hot functions of `dist-primitives`
//! recognise it through `MpcNet::hip_backend` (the one defaulted method `rust/patches/mpc-net.diff` adds to the trait)
//! and hand their round to the device; everything else (`client_send_or_king_receive_serialized` of small values in
//! code that was not patched) travels as host messages through the two collectives below.
use core::any::Any;
use core::ffi::c_void;
use core::ptr;
use std::collections::HashMap;
use std::time::Duration;

use async_trait::async_trait;
use mpc_net::{ClientSendOrKingReceiveResult, MpcNet, MpcNetError, MultiplexedStreamID};
use tokio_util::bytes::Bytes;
use zksaas_hip_sys as sys;

use crate::{check, Context, DeviceBuf};

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
    /// The `HipNet` behind any `Net: MpcNet` (through `&`, `&mut`, `Arc`: `auto_impl` forwards `hip_backend`), or
    /// `NotConnected` when the net is one of the reference's TCP nets.
    pub fn of<Net: MpcNet + ?Sized>(net: &Net) -> Result<&HipNet, MpcNetError> {
        net.hip_backend().and_then(|a| a.downcast_ref::<HipNet>()).ok_or(MpcNetError::NotConnected)
    }
    pub fn ctx(&self) -> &Context {
        &self.ctx
    }
    pub fn raw_net(&self) -> *mut sys::ZkNet {
        self.raw
    }
    /// parties driven by this rank (1 in the reference's one-process-per-party deployment)
    pub fn parties_per_rank(&self) -> usize {
        self.k
    }
    /// The round timeout (`lib.rs:98-135`, 30 s by default there and here).
    pub fn set_timeout_ms(&self, ms: u64) -> Result<(), MpcNetError> {
        check(&self.ctx, unsafe { sys::zk_net_set_timeout_ms(self.raw, ms) })
    }
    /// Waits for a channel's transfers with the timeout as a deadline (a hung collective becomes `Protocol`).
    pub fn sync(&self, sid: MultiplexedStreamID) -> Result<(), MpcNetError> {
        check(&self.ctx, unsafe { sys::zk_net_sync(self.raw, sid as i32) })
    }
    fn byte_collectives_need_one_party_per_rank(&self) -> Result<(), MpcNetError> {
        if self.k != 1 {
            return Err(MpcNetError::BadInput {
                err: "HipNet: byte-level collectives address parties; run one party per rank (world = n) or use the primitives",
            });
        }
        Ok(())
    }
}
impl Drop for HipNet {
    fn drop(&mut self) {
        unsafe { sys::zk_net_destroy(self.raw) }
    }
}

#[async_trait]
impl MpcNet for HipNet {
    /// `lib.rs:69`.
    fn n_parties(&self) -> usize {
        self.ctx.n
    }
    /// `lib.rs:71`: the first party this rank drives (party 0 lives on rank 0, so `is_king` keeps its default).
    fn party_id(&self) -> u32 {
        self.first_party as u32
    }
    fn is_init(&self) -> bool {
        !self.raw.is_null()
    }
    /// The defaulted method of `rust/patches/mpc-net.diff`.
    fn hip_backend(&self) -> Option<&(dyn Any + Send + Sync)> {
        Some(self)
    }
    /// The star has no point-to-point verb on its data plane (`zk_net_*` are collectives); the reference's hot path
    /// only reaches `send_to` / `recv_from` through the two collectives overridden below.
    async fn recv_from(&self, _id: u32, _sid: MultiplexedStreamID) -> Result<Bytes, MpcNetError> {
        Err(MpcNetError::BadInput { err: "HipNet: use client_send_or_king_receive / client_receive_or_king_send" })
    }
    async fn send_to(&self, _id: u32, _bytes: Bytes, _sid: MultiplexedStreamID) -> Result<(), MpcNetError> {
        Err(MpcNetError::BadInput { err: "HipNet: use client_send_or_king_receive / client_receive_or_king_send" })
    }
    /// `lib.rs:89-136` as one gather: `zk_net_enter` is the round's admission (the king waits up to the net's timeout
    /// and publishes the mask of present ranks), `zk_net_gather` moves the equal-length payloads through device
    /// staging buffers (RCCL over xGMI; the `*_host` verbs carry at most 4 KiB).  All present -> `Full`, otherwise
    /// `Partial` keyed by party -- what `ser_net.rs:35-94` expects.
    async fn client_send_or_king_receive(&self, bytes: &[u8], sid: MultiplexedStreamID, timeout: Duration)
                                         -> Result<Option<ClientSendOrKingReceiveResult>, MpcNetError> {
        self.byte_collectives_need_one_party_per_rank()?;
        self.set_timeout_ms(timeout.as_millis() as u64)?;
        let mut mask = 0u32;
        check(&self.ctx, unsafe { sys::zk_net_enter(self.raw, sid as i32, &mut mask) })?;
        let mine = DeviceBuf::from_slice(&self.ctx, bytes)?;
        let full = DeviceBuf::alloc(&self.ctx, if self.is_king() { self.world * bytes.len() } else { 1 })?;
        check(&self.ctx, unsafe {
            sys::zk_net_gather(self.raw, sid as i32, mask, mine.ptr(), bytes.len(), full.ptr())
        })?;
        self.sync(sid)?;
        if !self.is_king() {
            return Ok(None);
        }
        // the present ranks' blocks arrive compacted in rank order (include/zksaas.h, "raw verbs")
        let present: Vec<usize> = (0..self.world).filter(|r| mask >> r & 1 == 1).collect();
        let all: Vec<u8> = full.to_vec(present.len() * bytes.len())?;
        let part = |i: usize| Bytes::copy_from_slice(&all[i * bytes.len()..(i + 1) * bytes.len()]);
        if present.len() == self.world {
            Ok(Some(ClientSendOrKingReceiveResult::Full((0..self.world).map(part).collect())))
        } else {
            let got: HashMap<u32, Bytes> = present.iter().enumerate().map(|(i, r)| (*r as u32, part(i))).collect();
            Ok(Some(ClientSendOrKingReceiveResult::Partial(got)))
        }
    }
    /// `lib.rs:139-176`: the king's `n` equal-length answers are one scatter (rank `r` receives block `r`); the length
    /// goes first as a host message.  The equal-length check and its `Protocol` error are the reference's.
    async fn client_receive_or_king_send(&self, bytes_out: Option<Vec<Bytes>>, sid: MultiplexedStreamID)
                                         -> Result<Bytes, MpcNetError> {
        self.byte_collectives_need_one_party_per_rank()?;
        if bytes_out.is_some() != self.is_king() {
            return Err(MpcNetError::BadInput {
                err: if self.is_king() { "recv_from_king called with no bytes_out when king" }
                     else { "recv_from_king called with bytes_out when not king" },
            });
        }
        let mut mask = 0u32;
        check(&self.ctx, unsafe { sys::zk_net_enter(self.raw, sid as i32, &mut mask) })?;
        let mut len = [0u64; 1];
        let mut flat: Vec<u8> = Vec::new();
        if let Some(out) = &bytes_out {
            let m = out[0].len();
            if let Some(id) = (0..self.n_parties()).find(|id| out[*id].len() != m) {
                return Err(MpcNetError::Protocol { err: format!("Peer {} sent wrong number of bytes", id), party: id as u32 });
            }
            len[0] = m as u64;
            for b in out {
                flat.extend_from_slice(b);
            }
        }
        check(&self.ctx, unsafe { sys::zk_net_bcast_host(self.raw, sid as i32, mask, len.as_mut_ptr() as *mut c_void, 8) })?;
        let m = len[0] as usize;
        let full = if self.is_king() { DeviceBuf::from_slice(&self.ctx, &flat)? } else { DeviceBuf::alloc(&self.ctx, 1)? };
        let local = DeviceBuf::alloc(&self.ctx, m)?;
        check(&self.ctx, unsafe { sys::zk_net_scatter(self.raw, sid as i32, mask, full.ptr(), m, local.ptr()) })?;
        self.sync(sid)?;
        Ok(Bytes::from(local.to_vec::<u8>(m)?))
    }
}
