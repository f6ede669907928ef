-- NOT COMPILED IN THIS REPOSITORY'S IMAGE (no GHC here): this is the binding a maintainer of
-- jinilover/floydWarshall would add, kept as a file so that it can be dropped into src/lib/.
-- See INTEGRATION.md.  cabal: extra-libraries: fwx, build-depends: vector.
{-# LANGUAGE ForeignFunctionInterface #-}
-- Lazy variant (INTEGRATION.md section 2b): the solved matrix stays in HBM, entries are fetched by
-- fwx_matrix_query_exact when `optimum` forces them.
module AlgorithmsFFILazy (runAlgoGPULazy) where

import           Control.Monad                (when)
import           Data.Int                     (Int32)
import qualified Data.Vector                  as V
import qualified Data.Vector.Storable         as S
import           Foreign.C.Types              (CInt (..))
import           Foreign.Concurrent           (newForeignPtr)
import           Foreign.ForeignPtr           (withForeignPtr)
import           Foreign.Marshal.Alloc        (alloca)
import           Foreign.Marshal.Array        (allocaArray, peekArray)
import           Foreign.Ptr                  (Ptr, nullPtr)
import           Foreign.Storable             (peek)
import           System.IO.Unsafe             (unsafePerformIO)

import           Types                        (Matrix, RateEntry (..))

data FwxMatrix                                   -- opaque fwx_matrix
foreign import ccall safe "fwx.h fwx_matrix_create"
  c_create  :: Ptr (Ptr FwxMatrix) -> Int32 -> Int32 -> Int32 -> Int32 -> Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_enable_path_log" c_enable_log :: Ptr FwxMatrix -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_upload"
  c_upload  :: Ptr FwxMatrix -> Ptr Double -> Ptr Int32 -> Ptr Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_solve"   c_solve :: Ptr FwxMatrix -> Ptr () -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_query_exact"
  c_query   :: Ptr FwxMatrix -> Int32 -> Int32 -> Ptr Double -> Ptr Int32 -> Int32 -> IO CInt
-- int fwx_matrix_destroy(fwx_matrix*): the status is an int, so it is called (and its result
-- dropped) from a Haskell finalizer rather than passed as a `FunPtr (Ptr a -> IO ())`
foreign import ccall safe "fwx.h fwx_matrix_destroy" c_destroy :: Ptr FwxMatrix -> IO CInt

fwxErrCapacity :: CInt
fwxErrCapacity = -6                               -- FWX_ERR_CAPACITY (fwx.h)

-- | floydWarshall with the solved matrix left on the device (dtype 1 = f64, next-hops carried).
runAlgoGPULazy :: Matrix RateEntry -> Matrix RateEntry
runAlgoGPULazy m
  | n == 0    = m
  | otherwise = unsafePerformIO $ do
      h <- alloca $ \pp -> do { ok "create" =<< c_create pp (fromIntegral n) 1 1 0 (-1); peek pp }  -- f64, next, no hops
      fp <- newForeignPtr h (c_destroy h >> return ())   -- freed when the last entry thunk is dropped
      ok "enable_path_log" =<< c_enable_log h            -- path trace: exact `_path` lists under ties
      -- marshal rate / next exactly as runAlgoGPU does (row-major, head-of-path index or -1)
      let rate = S.fromList [ _bestRate e | row <- V.toList m, e <- V.toList row ]
          next = S.fromList [ headIdx e   | row <- V.toList m, e <- V.toList row ] :: S.Vector Int32
      S.unsafeWith rate $ \pr -> S.unsafeWith next $ \pn ->
        ok "upload" =<< c_upload h pr pn nullPtr          -- the library copies: nothing is retained
      ok "solve" =<< c_solve h nullPtr                    -- runAlgo 0, on the GPU
      -- (no hops: `length _path` is the length of the list query_exact returns, and without
      --  them the solve carries one n x n array less through every pass)
      return (V.generate n (\i -> V.generate n (\j -> entry fp i j)))
  where
    n = V.length m
    vertices = V.map (_start . V.head) m
    vIdx v   = maybe (-1) fromIntegral (V.elemIndex v vertices)
    headIdx e = case _path e of { [] -> -1; (v:_) -> vIdx v }
    ok what rc = when (rc /= 0) $ error ("libfwx " ++ what ++ ": status " ++ show rc)
    entry fp i j = unsafePerformIO $ withForeignPtr fp $ \h -> query h i j (max 64 (4 * n))
    -- forced by `optimum` only.  A negative status is an ERROR, never an empty path (an empty
    -- path means "There is no exchange", Algorithms.hs:75); FWX_ERR_CAPACITY is retried with a
    -- larger buffer, as Session::find_best_rate does (arbitrage inputs repeat vertices).
    query h i j cap =
      alloca $ \pr -> allocaArray cap $ \pp -> do
        len <- c_query h (fromIntegral i) (fromIntegral j) pr pp (fromIntegral cap)
        if len == fwxErrCapacity && cap < 16777216 then query h i j (cap * 8)
        else if len < 0 then error ("libfwx query_exact: status " ++ show len)
        else do
          r  <- peek pr
          ix <- peekArray (fromIntegral len) pp
          return (RateEntry r (vertices V.! i) [ vertices V.! fromIntegral x | x <- ix ])
