-- NOT COMPILED IN THIS REPOSITORY'S IMAGE (no GHC here): this is the binding a maintainer of
-- jinilover/floydWarshall would add, kept as a file so that it can be dropped into src/lib/.
-- See INTEGRATION.md.  cabal: extra-libraries: fwx, build-depends: vector.
{-# LANGUAGE ForeignFunctionInterface #-}
-- Lazy variant (INTEGRATION.md section 2b): the solved matrix stays in HBM, entries are fetched by
-- fwx_matrix_query_exact when `optimum` forces them.
module AlgorithmsFFILazy (runAlgoGPULazy) where

import           Data.Int                     (Int32)
import qualified Data.Vector                  as V
import           Foreign.C.Types              (CInt (..))
import           Foreign.ForeignPtr           (newForeignPtr, withForeignPtr)
import           Foreign.Marshal.Alloc        (alloca)
import           Foreign.Marshal.Array        (allocaArray, peekArray)
import           Foreign.Ptr                  (FunPtr, Ptr, nullPtr)
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
foreign import ccall "fwx.h &fwx_matrix_destroy"    p_destroy :: FunPtr (Ptr FwxMatrix -> IO ())

-- | floydWarshall with the solved matrix left on the device (dtype 1 = f64, next-hops carried).
runAlgoGPULazy :: Matrix RateEntry -> Matrix RateEntry
runAlgoGPULazy m
  | n == 0    = m
  | otherwise = unsafePerformIO $ do
      h <- alloca $ \pp -> do { ok =<< c_create pp (fromIntegral n) 1 1 0 (-1); peek pp }  -- f64, next, no hops
      fp <- newForeignPtr p_destroy h            -- freed when the last entry thunk is dropped
      ok =<< c_enable_log h                      -- path trace: exact `_path` lists under ties
      -- marshal rate / next exactly as in runAlgoGPU, then:
      --   ok =<< c_upload h pr pn nullPtr;  ok =<< c_solve h nullPtr
      -- (no hops: `length _path` is the length of the list query_exact returns, and without
      --  them a matrix of 256+ vertices takes the fused engine)
      return (V.generate n (\i -> V.generate n (\j -> entry fp i j)))
  where
    n = V.length m
    vertices = V.map (_start . V.head) m
    ok rc = if rc /= 0 then error ("libfwx: " ++ show rc) else return ()
    entry fp i j = unsafePerformIO $ withForeignPtr fp $ \h ->   -- forced by `optimum` only
      alloca $ \pr -> allocaArray cap $ \pp -> do
        len <- c_query h (fromIntegral i) (fromIntegral j) pr pp (fromIntegral cap)
        r   <- peek pr
        ix  <- peekArray (max 0 (fromIntegral len)) pp
        return (RateEntry r (vertices V.! i) [ vertices V.! fromIntegral x | x <- ix ])
    cap = 4 * n
