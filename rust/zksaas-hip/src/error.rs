//! `zk_status` + `zk_last_error` -> `mpc_net::MpcNetError` (`mpc-net/src/lib.rs:19-24`), variant for variant.
use core::ffi::CStr;

use mpc_net::MpcNetError;
use zksaas_hip_sys as sys;

use crate::Context;

fn message(ctx: &Context) -> (String, i32) {
    let mut party: i32 = -1;
    let p = unsafe { sys::zk_last_error(ctx.raw(), &mut party) };
    let msg = if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() };
    (msg, party)
}

/// `Ok(())` for `ZK_OK`, otherwise the reference's error with the library's message.
/// `BadInput { err: &'static str }` wants a static string: the library's text goes to the log, the variant carries
/// a fixed one (the reference's own `BadInput` sites are static messages too).
pub fn check(ctx: &Context, rc: i32) -> Result<(), MpcNetError> {
    match rc {
        sys::ZK_OK => Ok(()),
        sys::ZK_ERR_PROTOCOL => {
            let (err, party) = message(ctx);
            Err(MpcNetError::Protocol { err, party: party.max(0) as u32 })
        }
        sys::ZK_ERR_NOT_CONNECTED => Err(MpcNetError::NotConnected),
        sys::ZK_ERR_BAD_INPUT => {
            let (err, _) = message(ctx);
            eprintln!("zksaas-hip: bad input: {err}");
            Err(MpcNetError::BadInput { err: "zksaas-hip: bad input (see log)" })
        }
        _ => Err(MpcNetError::Generic(message(ctx).0)),
    }
}
